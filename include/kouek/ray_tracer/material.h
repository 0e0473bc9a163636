// Phong material and area lights with the reference's layout and factory functions
// (reference include/ray_tracer/material.h:12-99). BRDF evaluation and light sampling run on the GPU
// (raytracer_amd/csrc/path_trace.hip); the host side only builds the structs.
#ifndef KOUEK_RAY_TRACER_MATERIAL_H
#define KOUEK_RAY_TRACER_MATERIAL_H

#include <cmath>

#include "../hip/device.h"

namespace kouek {
namespace RayTracer {

struct alignas(16) Material {
    bool emissive;
    vec3 emission;
    vec3 kd;
    vec3 ks;
    vec3 tr;
    float ni;
    float ns;
    float kdOverKdAddKs;

    static Material Create(const vec3 &kd, const vec3 &ks, const vec3 &tr, float ni, float ns,
                           const vec3 &emission = vec3(0.f)) {
        Material matr{};
        matr.emissive = emission != vec3(0.f);
        matr.emission = emission;
        matr.kd = kd;
        matr.ks = ks;
        matr.tr = tr;
        matr.ns = ns;
        matr.ni = ni;
        auto kdAvg = (kd[0] + kd[1] + kd[2]) / 3.f;
        auto ksAvg = (ks[0] + ks[1] + ks[2]) / 3.f;
        matr.kdOverKdAddKs = kdAvg / (kdAvg + ksAvg);
        return matr;
    }
};

struct alignas(16) Light {
    enum class Type : uint8_t { Quad, Sphere };
    Type type;
    union {
        struct {
            vec3 o, u, v, n;
            float invArea;
        } quad;
        struct {
            vec3 o;
            float r, rSqr;
        } sphere;
    };
    float area;
    vec3 radiance;

    Light() : type(Type::Quad), quad{}, area(0.f), radiance() {}

    static Light CreateQuad(const vec3 &o, const vec3 &u, const vec3 &v, const vec3 &radiance) {
        Light lht;
        lht.type = Type::Quad;
        lht.quad.o = o;
        lht.quad.u = u;
        lht.quad.v = v;
        vec3 c(u[1] * v[2] - v[1] * u[2], u[2] * v[0] - v[2] * u[0], u[0] * v[1] - v[0] * u[1]);
        float inv = 1.f / std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        lht.quad.n = vec3(c[0] * inv, c[1] * inv, c[2] * inv);
        lht.area = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) *
                   std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        lht.quad.invArea = 1.f / lht.area;
        lht.radiance = radiance;
        return lht;
    }
    static Light CreateSphere(float r, const vec3 &o, const vec3 &radiance) {
        Light lht;
        lht.type = Type::Sphere;
        lht.sphere.o = o;
        lht.sphere.r = r;
        lht.sphere.rSqr = r * r;
        lht.area = 2.f * 6.28318530717958647692528676655900576f * lht.sphere.rSqr;
        lht.radiance = radiance;
        return lht;
    }
};
static_assert(sizeof(Material) == sizeof(kk_material) && sizeof(Light) == sizeof(kk_light),
              "layouts must match the C ABI");

} // namespace RayTracer
} // namespace kouek

#endif // KOUEK_RAY_TRACER_MATERIAL_H
