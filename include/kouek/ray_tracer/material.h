// Phong material and area lights with the reference's layout, factory functions, BRDF and light
// sampling (reference include/ray_tracer/material.h:12-139, helpers of include/cg/math.h:95-129). The
// renderer evaluates the same functions on the GPU (raytracer_amd/csrc/path_trace.hip); the host-callable
// members are kept because they are public members of the reference's structs. Their arithmetic is
// written component by component in glm's operation order, so they work with either vector type.
#ifndef KOUEK_RAY_TRACER_MATERIAL_H
#define KOUEK_RAY_TRACER_MATERIAL_H

#include <cmath>

#include "../hip/device.h"

namespace kouek {
namespace Math {
namespace detail {
inline vec3 v3(float a, float b, float c) { return vec3(a, b, c); }
inline float dot3(const vec3 &a, const vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline vec3 cross3(const vec3 &x, const vec3 &y) {
    return v3(x[1] * y[2] - y[1] * x[2], x[2] * y[0] - y[2] * x[0], x[0] * y[1] - y[0] * x[1]);
}
inline vec3 normalize3(const vec3 &v) { // glm: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
    const float s = 1.f / std::sqrt(dot3(v, v));
    return v3(v[0] * s, v[1] * s, v[2] * s);
}
inline vec3 frameToWorld(const vec3 &dir, const vec3 &norm, const vec3 &tngnt) {
    const vec3 bi = normalize3(cross3(norm, tngnt)); // glm::mat3(tngnt, biTngnt, norm) * dir
    return normalize3(v3(tngnt[0] * dir[0] + bi[0] * dir[1] + norm[0] * dir[2],
                         tngnt[1] * dir[0] + bi[1] * dir[1] + norm[1] * dir[2],
                         tngnt[2] * dir[0] + bi[2] * dir[1] + norm[2] * dir[2]));
}
} // namespace detail

// include/cg/math.h:95-129
inline vec3 GenerateTangent(const vec3 &norm) {
    if (std::fabs(norm[1]) < .999f)
        return detail::normalize3(detail::cross3(vec3(0.f, 1.f, 0.f), norm));
    return detail::normalize3(detail::cross3(norm, vec3(1.f, 0.f, 0.f)));
}
inline vec3 ThetaPhiToDirection(float theta, float phi, const vec3 &norm, const vec3 &tngnt) {
    const float sinTheta = std::sin(theta);
    return detail::frameToWorld(vec3(std::cos(phi) * sinTheta, std::sin(phi) * sinTheta, std::cos(theta)),
                                norm, tngnt);
}
inline vec3 CosineThetaPhiToDirection(float cosTheta, float phi, const vec3 &norm, const vec3 &tngnt) {
    float s = 1.f - cosTheta * cosTheta;
    s = s < 0.f ? 0.f : (s > 1.f ? 1.f : s); // glm::clamp = min(max(x, lo), hi)
    const float sinTheta = std::sqrt(s);
    return detail::frameToWorld(vec3(std::cos(phi) * sinTheta, std::sin(phi) * sinTheta, cosTheta), norm,
                                tngnt);
}
} // namespace Math

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

    // reference material.h:41-48
    vec3 BRDF(const vec3 &dirIn, const vec3 &dirOut, const vec3 &norm) const {
        using namespace Math::detail;
        constexpr float OneOverPi = 0.318309886183790671537767526745028724f;
        constexpr float OneOverTwoPi = 0.159154943091895335768883763372514362f;
        const float nd = 2.f * dot3(norm, dirOut);
        const vec3 reflect =
            normalize3(v3(nd * norm[0] - dirOut[0], nd * norm[1] - dirOut[1], nd * norm[2] - dirOut[2]));
        const float d = dot3(dirIn, reflect);
        const float lobe = std::pow(d < 0.f ? 0.f : d, ns); // glm::max(x, 0.f)
        vec3 out;
        for (int i = 0; i < 3; ++i)
            out[i] = kd[i] * OneOverPi + ks[i] * (ns + 2.f) * OneOverTwoPi * lobe;
        return out;
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

    // reference material.h:95-114 (uniform hemisphere sampling of the sphere, as the reference compiles it)
    struct SampleResult {
        vec3 pos;
        vec3 norm;
        float pdf;
    };
    SampleResult SampleQuad(float u0, float u1) const {
        SampleResult r;
        for (int i = 0; i < 3; ++i)
            r.pos[i] = quad.o[i] + u0 * quad.u[i] + u1 * quad.v[i];
        r.norm = quad.n;
        r.pdf = quad.invArea;
        return r;
    }
    SampleResult SampleSphere(float u0, float u1, const vec3 &pos) const {
        using namespace Math::detail;
        constexpr float TwoPi = 6.28318530717958647692528676655900576f;
        const vec3 dirO2P =
            normalize3(v3(pos[0] - sphere.o[0], pos[1] - sphere.o[1], pos[2] - sphere.o[2]));
        const vec3 tngnt = Math::GenerateTangent(dirO2P);
        SampleResult r;
        r.norm = Math::ThetaPhiToDirection(std::acos(1 - u0), TwoPi * u1, dirO2P, tngnt);
        for (int i = 0; i < 3; ++i)
            r.pos[i] = sphere.o[i] + sphere.r * r.norm[i];
        r.pdf = 0.159154943091895335768883763372514362f;
        return r;
    }
};
static_assert(sizeof(Material) == sizeof(kk_material) && sizeof(Light) == sizeof(kk_light),
              "layouts must match the C ABI");

} // namespace RayTracer
} // namespace kouek

#endif // KOUEK_RAY_TRACER_MATERIAL_H
