// Path-tracing frame: kouek::RayTracer::RayTracer::Render(..., RenderTarget::Scene)
// (src/ray_tracer/ray_tracer_render.cu:561-638) as one kernel, one lane per pixel:
//   ray generation (:592-619), renderScene (:312-431) with Material::BRDF and the light sampling of
//   include/ray_tracer/material.h:41-48,100-114 and the helpers of include/cg/math.h:75-129,
//   running-mean accumulation and RGBA8 output (:623-637).
// Closest hits use the walk of traverse.h over the build's own arrays or, when the scene names a
// trace-side tree, the 8-wide walk of wide_walk.h (same hits). Shadow rays only need "is anything
// nearer than the light" (:381-383 compares the closest t with the distance), so they stop at the
// first such triangle.
// Random numbers: the reference draws from cuRAND XORWOW, curand_init(1234, pixel, 0); cuRAND does not
// exist on this platform, so a counter-based Philox4x32-10 keyed by (seed; draw, sample, pixel) is used,
// drawn in the reference's order, which also makes a pixel's stream independent of how the image is
// split between GPUs.
#include "kk_host.h"
#include "traverse.h"
#include "wide_walk.h"

namespace kk {
namespace pt {

constexpr int TILE_W = 16, TILE_H = 16;
constexpr int THREADS = TILE_W * TILE_H;
constexpr float Eps = .001f; // include/ray_tracer/ray_tracer.h:27
constexpr float TwoPi = 6.28318530717958647692528676655900576f;
constexpr float OneOverPi = 0.318309886183790671537767526745028724f;
constexpr float OneOverTwoPi = 0.159154943091895335768883763372514362f;

struct Rng {
    uint32_t seed, pixel, sample, draw;
    KK_D float next() {
        uint32_t c0 = draw++, c1 = sample, c2 = pixel, c3 = 0x4B4F5545u;
        uint32_t k0 = seed, k1 = 0xCAFEF00Du;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            const uint32_t n0 = hi1 ^ c1 ^ k0;
            const uint32_t n2 = hi0 ^ c3 ^ k1;
            c0 = n0;
            c1 = lo1;
            c2 = n2;
            c3 = lo0;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        return static_cast<float>((c0 >> 8) + 1u) * 5.9604644775390625e-08f; // (0, 1]
    }
};

KK_D f3 V(const float *p) { return {p[0], p[1], p[2]}; }

// include/cg/math.h:95-129
KK_D f3 GenerateTangent(f3 norm) {
    if (fabsf(norm.y) < .999f)
        return normalize(cross(F3(0.f, 1.f, 0.f), norm));
    return normalize(cross(norm, F3(1.f, 0.f, 0.f)));
}
KK_D f3 FrameToWorld(f3 dir, f3 norm, f3 tngnt) {
    const f3 biTngnt = normalize(cross(norm, tngnt));
    const m3 m{{tngnt, biTngnt, norm}};
    return normalize(m * dir);
}
KK_D f3 ThetaPhiToDirection(float theta, float phi, f3 norm, f3 tngnt) {
    const float sinTheta = sinf(theta);
    const f3 dir = {cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosf(theta)};
    return FrameToWorld(dir, norm, tngnt);
}
KK_D f3 CosineThetaPhiToDirection(float cosTheta, float phi, f3 norm, f3 tngnt) {
    const float sinTheta = sqrtf(gmin(gmax(1.f - cosTheta * cosTheta, 0.f), 1.f));
    const f3 dir = {cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta};
    return FrameToWorld(dir, norm, tngnt);
}

// include/ray_tracer/material.h:41-48
KK_D f3 BRDF(const kk_material &m, f3 dirIn, f3 dirOut, f3 norm) {
    const f3 diffuse = V(m.kd) * OneOverPi;
    const f3 reflect = normalize(2.f * dot(norm, dirOut) * norm - dirOut);
    const f3 specular =
        V(m.ks) * (m.ns + 2.f) * OneOverTwoPi * powf(gmax(dot(dirIn, reflect), 0.f), m.ns);
    return diffuse + specular;
}

struct LightSample {
    f3 pos, norm;
    float pdf;
};
// include/ray_tracer/material.h:100-114
KK_D LightSample SampleLight(const kk_light &l, float u0, float u1, f3 pos) {
    LightSample r;
    if (l.type == 0) {
        r.pos = V(l.quad.o) + u0 * V(l.quad.u) + u1 * V(l.quad.v);
        r.norm = V(l.quad.n);
        r.pdf = l.quad.inv_area;
        return r;
    }
    const f3 dirO2P = normalize(pos - V(l.sphere.o));
    const f3 tngnt = GenerateTangent(dirO2P);
    r.norm = ThetaPhiToDirection(acosf(1 - u0), TwoPi * u1, dirO2P, tngnt);
    r.pos = V(l.sphere.o) + l.sphere.r * r.norm;
    r.pdf = OneOverTwoPi;
    return r;
}

// True when some triangle is hit at 0 <= t < tMax, i.e. when the closest hit is nearer than tMax.
KK_D bool occluded(const trav::Scene &s, const Ray &ray, float tMax, uint32_t *stack, int stride) {
    using namespace trav;
    const NodeData root = load_node(s.nodes, 0);
    float tEnter, tExit;
    HitAABB(ray, root.box, tEnter, tExit);
    if (tEnter > tExit)
        return false;
    uint32_t lft = root.lft, rht = root.rht;
    int sp = 0;
    const float limit = tMax + CULL_SLACK * tMax;
    while (true) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const uint32_t ch = side == 0 ? lft : rht;
            if ((ch & LEAF) && (side == 0 || ch != lft)) {
                const uint32_t c = ch & LEAF_MASK;
                uint32_t k = s.compacted[c];
                const uint32_t kEnd = (c == s.compactedNum - 1) ? s.faceNum : s.compacted[c + 1];
                for (; k < kEnd; ++k) {
                    const uint32_t fi = s.faces[k];
                    float bx, by;
                    const float t = HitTriangle(ray, load_pos(s.positions, s.tri[3ull * fi]),
                                                load_pos(s.positions, s.tri[3ull * fi + 1]),
                                                load_pos(s.positions, s.tri[3ull * fi + 2]), bx, by);
                    if (t >= 0.f && t < tMax)
                        return true;
                }
            }
        }
        bool hitL = false, hitR = false;
        NodeData nl, nr;
        if (!(lft & LEAF)) {
            nl = load_node(s.nodes, lft);
            HitAABB(ray, nl.box, tEnter, tExit);
            hitL = !(tEnter > tExit) && !(tEnter > limit);
        }
        if (!(rht & LEAF)) {
            nr = load_node(s.nodes, rht);
            HitAABB(ray, nr.box, tEnter, tExit);
            hitR = !(tEnter > tExit) && !(tEnter > limit);
        }
        if (hitL && hitR && sp < STACK_DEPTH) {
            stack[sp * stride] = rht;
            ++sp;
            lft = nl.lft;
            rht = nl.rht;
        } else if (hitL) {
            lft = nl.lft;
            rht = nl.rht;
        } else if (hitR) {
            lft = nr.lft;
            rht = nr.rht;
        } else if (sp > 0) {
            --sp;
            const uint2 ch = *reinterpret_cast<const uint2 *>(s.nodes + stack[sp * stride]);
            lft = ch.x;
            rht = ch.y;
        } else
            return false;
    }
}

// The two ways a lane finds hits: the binary tree as built, or the 8-wide trace-side tree.
struct BinaryTracer {
    trav::Scene s;
    uint32_t *stack;
    int stride;
    KK_D void closest(const Ray &ray, float &t, float &bx, float &by, uint32_t &fi) const {
        uint32_t a = 0, b = 0;
        const trav::Hit h = trav::closest_hit<false>(s, ray, stack, stride, a, b);
        t = h.t, bx = h.bx, by = h.by, fi = h.fi;
    }
    KK_D bool shadowed(const Ray &ray, float tMax) const { return occluded(s, ray, tMax, stack, stride); }
};
struct WideTracer {
    accel::View v;
    accel::LaneStack st;
    KK_D void closest(const Ray &ray, float &t, float &bx, float &by, uint32_t &fi) const {
        const accel::Hit8 h = accel::closest_hit_lane(v, ray, st);
        t = h.t, bx = h.bx, by = h.by, fi = h.fi;
    }
    KK_D bool shadowed(const Ray &ray, float tMax) const { return accel::occluded_lane(v, ray, tMax, st); }
};

// src/ray_tracer/ray_tracer_render.cu:312-431
template <typename Tracer>
KK_D f3 renderScene(const Tracer &tracer, const kk_pt_scene &p, Rng &rng, Ray ray) {
    constexpr float RR = .8f;
    f3 throughout = F3(1.f);
    f3 Lo = F3(0.f);
    for (uint8_t depth = 0;
         depth < p.max_path_depth && gmax(throughout.x, gmax(throughout.y, throughout.z)) >= FloatEps;
         ++depth) {
        struct {
            float t, bx, by;
            uint32_t fi;
        } nearest;
        tracer.closest(ray, nearest.t, nearest.bx, nearest.by, nearest.fi);
        if (nearest.t == FloatMax)
            break;
        const kk_pt_triangle triangle = p.triangles[nearest.fi];
        const kk_material matr = p.materials[triangle.grp_idx];
        if (matr.emissive) {
            Lo = Lo + throughout * V(matr.emission);
            break;
        }
        const f3 norm = normalize(nearest.bx * trav::load_pos(p.normals, triangle.norm_idx[0]) +
                                  nearest.by * trav::load_pos(p.normals, triangle.norm_idx[1]) +
                                  (1.f - nearest.bx - nearest.by) *
                                      trav::load_pos(p.normals, triangle.norm_idx[2]));
        const f3 dirOut = -ray.dir;
        ray.pos = ray.pos + nearest.t * ray.dir + Eps * norm;

        f3 Li = F3(0.f);
        for (uint32_t li = 0; li < p.light_num; ++li) {
            const kk_light lht = p.lights[li];
            const float u0 = rng.next();
            const float u1 = rng.next();
            LightSample ls = SampleLight(lht, u0, u1, ray.pos);
            ls.pos = ls.pos + Eps * ls.norm;
            const f3 in = ls.pos - ray.pos;
            const float sqrIn = dot(in, in);
            ray.dir = normalize(in);
            const float factor = gmax(dot(ray.dir, norm), 0.f) * gmax(dot(-ray.dir, ls.norm), 0.f) /
                                 sqrIn * lht.area / p.light_tot_area;
            if (factor < FloatEps)
                continue;
            if (tracer.shadowed(ray, sqrtf(sqrIn) - Eps))
                continue;
            Li = Li + factor / ls.pdf * V(lht.radiance) * BRDF(matr, ray.dir, dirOut, norm);
        }

        float pdfIndrct;
        float u = rng.next();
        const float u0 = rng.next();
        const float u1 = rng.next();
        if (u <= matr.kd_over_kd_add_ks) {
            const f3 tngnt = GenerateTangent(norm);
            ray.dir = CosineThetaPhiToDirection(sqrtf(1.f - u0), TwoPi * u1, norm, tngnt);
            pdfIndrct = dot(ray.dir, norm) * OneOverPi * matr.kd_over_kd_add_ks;
        } else {
            const f3 reflect = normalize(2.f * dot(norm, dirOut) * norm - dirOut);
            const f3 tngnt = GenerateTangent(reflect);
            ray.dir = ThetaPhiToDirection(acosf(powf(u0, 1.f / (matr.ns + 1.f))), TwoPi * u1, reflect,
                                          tngnt);
            pdfIndrct = (matr.ns + 1.f) * OneOverTwoPi *
                        powf(gmax(dot(reflect, ray.dir), 0.f), matr.ns) *
                        (1.f - matr.kd_over_kd_add_ks);
        }
        Lo = Lo + throughout * Li;
        throughout = throughout * (dot(ray.dir, norm) / pdfIndrct / RR * BRDF(matr, ray.dir, dirOut, norm));

        u = rng.next();
        if (u > RR && depth > 1)
            break;
    }
    // HDRToLDRCorrect + GammaCorrect, include/cg/math.h:75-93
    Lo = Lo / (Lo + 1.f);
    Lo.x = powf(Lo.x, 1.f / 2.2f);
    Lo.y = powf(Lo.y, 1.f / 2.2f);
    Lo.z = powf(Lo.z, 1.f / 2.2f);
    return Lo;
}

template <bool WIDE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) void path_trace_kernel(
    const kk_lbvh_device_data *__restrict__ lbvh, const accel::View view, const kk_pt_scene p,
    const kk_pt_frame f, float *__restrict__ accum, uint8_t *__restrict__ rgba, int W, int H, int y0, int y1,
    float sampleIdx, uint32_t seed) {
    // binary walk: one word per entry; wide walk: a group (base word + 16 bits) per entry
    constexpr int WORDS = WIDE ? accel::LANE_LDS_DEPTH : trav::STACK_DEPTH;
    __shared__ uint32_t s_stack[WORDS * THREADS];
    __shared__ unsigned short s_bits[WIDE ? accel::LANE_LDS_DEPTH * THREADS : 1];
    const int tid = threadIdx.x;
    // a wave covers a 16 x 4 pixel patch of the tile
    const int x = blockIdx.x * TILE_W + (tid & (TILE_W - 1));
    const int y = y0 + blockIdx.y * TILE_H + (tid / TILE_W);
    if (x >= W || y >= y1)
        return;

    const uint32_t idx = static_cast<uint32_t>(y) * W + x;
    Rng rng{seed, idx, static_cast<uint32_t>(sampleIdx), 0};
    Ray eyeRay;
    {
        m4 invProj, w2s;
        m3 rot;
        for (int c = 0; c < 4; ++c) {
            invProj.c[c] = {p.inv_proj[4 * c], p.inv_proj[4 * c + 1], p.inv_proj[4 * c + 2],
                            p.inv_proj[4 * c + 3]};
            w2s.c[c] = {p.w2s[4 * c], p.w2s[4 * c + 1], p.w2s[4 * c + 2], p.w2s[4 * c + 3]};
        }
        for (int c = 0; c < 3; ++c)
            rot.c[c] = {f.eye_rot[3 * c], f.eye_rot[3 * c + 1], f.eye_rot[3 * c + 2]};
        f4 tmp;
        tmp.x = 1.f - rng.next() + x;
        tmp.y = 1.f - rng.next() + y;
        tmp.z = static_cast<float>(W);
        tmp.w = static_cast<float>(H);
        tmp.x = (2.f * tmp.x - tmp.z) / tmp.z;
        tmp.y = (2.f * tmp.y - tmp.w) / tmp.w;
        tmp.z = 1.f;
        tmp.w = 1.f;
        tmp = invProj * tmp;
        eyeRay.dir = {tmp.x, tmp.y, tmp.z};
        eyeRay.dir = rot * normalize(eyeRay.dir);
        // the reference transforms a temporary by w2s and discards it (:612-614)
        eyeRay.dir = normalize(eyeRay.dir);
        const f4 pp = w2s * f4{f.eye_pos[0], f.eye_pos[1], f.eye_pos[2], 1.f};
        eyeRay.pos = {pp.x, pp.y, pp.z};
    }
    f3 color;
    if (WIDE) {
        const WideTracer tracer{view, accel::LaneStack{s_stack + tid, s_bits + tid, THREADS}};
        color = renderScene(tracer, p, rng, eyeRay);
    } else {
        BinaryTracer tracer;
        tracer.s.nodes = lbvh->internal_nodes;
        tracer.s.faces = lbvh->faces;
        tracer.s.compacted = lbvh->compacted_faces;
        tracer.s.faceNum = lbvh->face_num;
        tracer.s.compactedNum = lbvh->compacted_face_num;
        tracer.s.positions = p.positions;
        tracer.s.tri = p.tri_to_position_indices;
        tracer.stack = s_stack + tid;
        tracer.stride = THREADS;
        color = renderScene(tracer, p, rng, eyeRay);
    }
    const f3 prev = {accum[3ull * idx], accum[3ull * idx + 1], accum[3ull * idx + 2]};
    color = prev + (color - prev) / sampleIdx;
    accum[3ull * idx] = color.x;
    accum[3ull * idx + 1] = color.y;
    accum[3ull * idx + 2] = color.z;
    uchar4 o;
    o.x = static_cast<uint8_t>(255.f * color.x);
    o.y = static_cast<uint8_t>(255.f * color.y);
    o.z = static_cast<uint8_t>(255.f * color.z);
    o.w = 255;
    reinterpret_cast<uchar4 *>(rgba)[idx] = o;
}

__global__ void rng_uniforms_kernel(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t count,
                                    float *out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Rng rng{seed, pixel, sample, 0};
        for (uint32_t i = 0; i < count; ++i)
            out[i] = rng.next();
    }
}

} // namespace pt
} // namespace kk

extern "C" int kk_pt_render_sample(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                                   const kk_pt_frame *h_frame, float *d_accum, uint8_t *d_rgba,
                                   int width, int height, int y0, int y1, float sample_index,
                                   uint32_t seed, void *stream_) {
    using namespace kk;
    using namespace kk::pt;
    KK_REQUIRE(d_lbvh && h_scene && h_frame && d_accum && d_rgba, "null pointer");
    KK_REQUIRE(width > 0 && height > 0 && y0 >= 0 && y1 <= height && y0 <= y1, "bad image size or rows");
    KK_REQUIRE(sample_index >= 1.f, "sample_index is 1-based");
    KK_REQUIRE(h_scene->triangles && h_scene->tri_to_position_indices && h_scene->positions &&
                   h_scene->normals && h_scene->materials,
               "scene arrays missing");
    KK_REQUIRE(h_scene->light_num == 0 || h_scene->lights, "lights missing");
    if (y0 == y1)
        return KK_OK;
    dim3 grid(div_up(width, TILE_W), div_up(y1 - y0, TILE_H));
    if (h_scene->trace_accel) {
        KK_REQUIRE(h_scene->accel_face_num > 0 && h_scene->accel_compacted_face_num > 0 &&
                       h_scene->accel_compacted_face_num <= h_scene->accel_face_num,
                   "trace accel sizes out of range");
        const accel::View view =
            accel::make_view(h_scene->trace_accel, h_scene->accel_face_num, h_scene->accel_compacted_face_num);
        path_trace_kernel<true><<<grid, THREADS, 0, static_cast<hipStream_t>(stream_)>>>(
            d_lbvh, view, *h_scene, *h_frame, d_accum, d_rgba, width, height, y0, y1, sample_index, seed);
    } else
        path_trace_kernel<false><<<grid, THREADS, 0, static_cast<hipStream_t>(stream_)>>>(
            d_lbvh, accel::View{}, *h_scene, *h_frame, d_accum, d_rgba, width, height, y0, y1, sample_index,
            seed);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

extern "C" int kk_rng_uniforms(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t count,
                               float *d_out, void *stream_) {
    KK_REQUIRE(d_out != nullptr, "null pointer");
    kk::pt::rng_uniforms_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream_)>>>(seed, pixel, sample,
                                                                                 count, d_out);
    KK_LAUNCH_CHECK();
    return KK_OK;
}
