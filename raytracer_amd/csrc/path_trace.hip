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
// exist on this platform, so a counter-based Philox4x32-10 keyed by (seed; event, sample, pixel) is used -
// one block of four uniforms per sampling event (pixel jitter, light sample, bounce), its words used in
// the reference's order of draws - which also makes a pixel's stream independent of how the image is
// split between GPUs.
#include "kk_host.h"
#include "traverse.h"
#include "wide_walk.h"

#include <atomic>
#include <mutex>

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
    // The next four uniforms in (0, 1]: the four words of one Philox4x32-10 block. A sampling event of a
    // path - the pixel jitter, a light sample, a bounce - takes one block and uses its words in the
    // reference's order of draws.
    KK_D void next4(float &u0, float &u1, float &u2, float &u3) {
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
        u0 = static_cast<float>((c0 >> 8) + 1u) * 5.9604644775390625e-08f;
        u1 = static_cast<float>((c1 >> 8) + 1u) * 5.9604644775390625e-08f;
        u2 = static_cast<float>((c2 >> 8) + 1u) * 5.9604644775390625e-08f;
        u3 = static_cast<float>((c3 >> 8) + 1u) * 5.9604644775390625e-08f;
    }
};

// x^y for x >= 0 where only a magnitude depends on it (the specular lobe's value and density, the tone curve):
// exp2(y log2 x) on the hardware's log2 and exp2, a few parts in a million off the library's powf against the
// 1e-3 the images are compared within, at four instructions instead of about a hundred. (x = 0 gives 0 for
// y > 0, y = 0 gives 1, as powf does.) The direction of a specular bounce keeps powf: a direction decides
// what a ray hits.
KK_D float magnitudePow(float x, float y) {
    const float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
    return y == 0.f ? 1.f : r;
}
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
// (sine and cosine of one angle from one call: the same values as sinf and cosf give, one argument reduction)
KK_D f3 ThetaPhiToDirection(float theta, float phi, f3 norm, f3 tngnt) {
    float sinTheta, cosTheta, sinPhi, cosPhi;
    sincosf(theta, &sinTheta, &cosTheta);
    sincosf(phi, &sinPhi, &cosPhi);
    const f3 dir = {cosPhi * sinTheta, sinPhi * sinTheta, cosTheta};
    return FrameToWorld(dir, norm, tngnt);
}
KK_D f3 CosineThetaPhiToDirection(float cosTheta, float phi, f3 norm, f3 tngnt) {
    const float sinTheta = sqrtf(gmin(gmax(1.f - cosTheta * cosTheta, 0.f), 1.f));
    float sinPhi, cosPhi;
    sincosf(phi, &sinPhi, &cosPhi);
    const f3 dir = {cosPhi * sinTheta, sinPhi * sinTheta, cosTheta};
    return FrameToWorld(dir, norm, tngnt);
}

// include/ray_tracer/material.h:41-48
KK_D f3 BRDF(const kk_material &m, f3 dirIn, f3 dirOut, f3 norm) {
    const f3 diffuse = V(m.kd) * OneOverPi;
    const f3 reflect = normalize(2.f * dot(norm, dirOut) * norm - dirOut);
    const f3 specular =
        V(m.ks) * (m.ns + 2.f) * OneOverTwoPi * magnitudePow(gmax(dot(dirIn, reflect), 0.f), m.ns);
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
    if (tEnter > tExit || !ray_is_finite(ray.pos, ray.dir))
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
            float u0, u1, unused2, unused3;
            rng.next4(u0, u1, unused2, unused3);
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
        float u, u0, u1, uRoulette;
        rng.next4(u, u0, u1, uRoulette);
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
                        magnitudePow(gmax(dot(reflect, ray.dir), 0.f), matr.ns) *
                        (1.f - matr.kd_over_kd_add_ks);
        }
        Lo = Lo + throughout * Li;
        throughout = throughout * (dot(ray.dir, norm) / pdfIndrct / RR * BRDF(matr, ray.dir, dirOut, norm));

        if (uRoulette > RR && depth > 1)
            break;
    }
    // HDRToLDRCorrect + GammaCorrect, include/cg/math.h:75-93
    Lo = Lo / (Lo + 1.f);
    Lo.x = magnitudePow(Lo.x, 1.f / 2.2f);
    Lo.y = magnitudePow(Lo.y, 1.f / 2.2f);
    Lo.z = magnitudePow(Lo.z, 1.f / 2.2f);
    return Lo;
}

#ifndef KK_PT_SHORT_ONLY
template <bool WIDE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) void path_trace_kernel(
    const kk_lbvh_device_data *__restrict__ lbvh, const accel::View view, const kk_pt_scene p,
    const kk_pt_frame f, float *__restrict__ accum, uint8_t *__restrict__ rgba, int W, int H, int y0, int y1,
    float sampleIdx, uint32_t seed, int outRowShift) {
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
        float jx, jy, unused2, unused3;
        rng.next4(jx, jy, unused2, unused3);
        tmp.x = 1.f - jx + x;
        tmp.y = 1.f - jy + y;
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
    // image row y is row y - outRowShift of the output arrays (0 for a whole image; a stripe of a compact
    // block sits outRowShift rows lower than in the image); the random stream stays keyed by the image pixel
    const uint32_t odx = idx - static_cast<uint32_t>(outRowShift) * W;
    const f3 prev = {accum[3ull * odx], accum[3ull * odx + 1], accum[3ull * odx + 2]};
    color = prev + (color - prev) / sampleIdx;
    accum[3ull * odx] = color.x;
    accum[3ull * odx + 1] = color.y;
    accum[3ull * odx + 2] = color.z;
    uchar4 o;
    o.x = static_cast<uint8_t>(255.f * color.x);
    o.y = static_cast<uint8_t>(255.f * color.y);
    o.z = static_cast<uint8_t>(255.f * color.z);
    o.w = 255;
    reinterpret_cast<uchar4 *>(rgba)[odx] = o;
}

// ---- the frame as a persistent, compacting kernel (8-wide tree) ----------------------------------------
//
// One lane per pixel and launch leaves a wave waiting for its longest path while the lanes of finished
// paths idle. Here a grid of resident wave64 workgroups pulls pixels from a queue (8 x 8 pixel tiles in
// row-major tile order) and every lane is a small state machine over the rays of its path:
//   walking a closest-hit ray | walking a shadow ray | walk over, to be shaded | idle.
// The wave alternates between two phases. Walk: in every pass each lane with a ray and no triangles
// pending visits one record, then the triangles all those visits left pending are tested as one pool, 64
// (ray, triangle) pairs at a time whichever lanes they belong to - a lane ends up with the hit of the
// per-lane sequence of closest_hit_lane / occluded_lane, and neither a long leaf nor a pass with few
// lanes at triangles costs the wave a pass per triangle. Shade: when no lane walks any more, every lane
// consumes its walk's result (hit -> material, normal; shadow answer -> add the light), samples lights
// until one needs a shadow ray, or samples the path's next direction, and lanes whose path ended write
// their pixel and take the next pixels of the queue with one ballot + one atomic per wave. Each of these
// sections runs at most once per phase, and only when half the wave waits for it (SECTION_MIN): the
// lanes of a wave are at different places of their paths, and a section run for a handful of lanes
// costs the wave as much as one run for all; the few wait a walk phase longer instead. A lane computes
// its path with the same operations in the same order as renderScene above, and the random stream is
// keyed by (pixel, sample, sampling event), so the image does not depend on the scheduling.
#endif // KK_PT_SHORT_ONLY
namespace persist {

enum : uint32_t {
    ST_IDLE = 0,       // no pixel (queue used up)
    ST_WALK_C = 1,     // walking a closest-hit ray
    ST_WALK_S = 2,     // walking a shadow ray
    ST_DONE_C = 3,     // closest-hit walk over: best holds the hit
    ST_DONE_S_HIT = 4, // shadow ray blocked
    ST_DONE_S_CLEAR = 5,
    ST_LIGHTS = 6, // choosing the next light that needs a shadow ray, or the path's next direction
    ST_END = 7,    // path over: pixel to be written
    ST_LAUNCH_C = 8, // a closest-hit ray (ray.pos, ray.dir) waits to be set up ...
    ST_LAUNCH_S = 9, // ... a shadow ray that ends at rayFar
    ST_NEXT_SAMPLE = 10 // has a pixel, the camera ray of its next sample is to be made
};

// Resident waves per SIMD = workgroups per CU, a compile-time choice of the kernel. Measured on veach-mis
// 1080p: a launch of one sample pass (1.5 ms) is fastest with 3 (168 registers, nothing spilled: 23.7 ms
// per 16 passes; 4 -> 24.1, 5 -> 29.5, 6 -> 27.3), a launch of 16 samples per pixel with 4 (21.5 ms
// against 22.3): the long launch gains more from the fourth wave than it loses to 62 spilled registers.
constexpr int WAVES_SHORT = 3, WAVES_LONG = 4;
constexpr int LONG_LAUNCH_SAMPLES = 4; // from this many samples per pixel on a launch counts as long
// A section of the shading code (light sample, next direction, pixel write + refill) runs for a wave when
// at least this many of its lanes wait for it (or as many as wait for the fullest of the three, if that
// is fewer), or when no lane of the wave has a ray left to walk.
// Measured on veach-mis 1080p: 1 -> 30.2 ms, 16 -> 26.2, 32 -> 24.8, 48 -> 28.3, 64 -> 30.9.
constexpr int SECTION_MIN = 32;
constexpr uint32_t COUNTER_SLOTS = 64;
// Queue heads of the launches in flight: a launch takes the next slot and clears it on its stream
// first. (Launches more than COUNTER_SLOTS apart share a slot; they cannot overlap unless that many
// are in flight on one GPU at once.)
// One word takes about 88 returning atomics per microsecond, which a frame's ~70 000 refills of a
// single-sample launch would use up: the pixels are handed out from PIXEL_QUEUES queues (every
// PIXEL_QUEUES-th tile each, so that all hold the same mix of cheap and costly pixels; a wave starts at
// the queue of its XCD and moves on when that is used up), their heads 128 bytes apart.
constexpr uint32_t PIXEL_QUEUES = 8, HEAD_STRIDE = 32, SLOT_WORDS = PIXEL_QUEUES * HEAD_STRIDE;
#ifndef KK_PT_SHORT_ONLY
static __device__ uint32_t g_pixelQueue[COUNTER_SLOTS * SLOT_WORDS];
#endif

struct PathState {
    // the path
    uint32_t pixel;  // y * W + x
    uint32_t sample; // which sample of the pixel this path is (1-based)
    uint32_t draw;   // random numbers drawn so far
    uint32_t depth;  // bounce
    uint32_t li;     // light being sampled
    uint32_t grp;    // material of the last hit
    float fOverPdf;  // geometric factor / pdf of the light sample whose shadow ray is out
    f3 throughout, Lo, Li, norm, dirOut;
};

} // namespace persist

// STRIPES: the rows [y0, y1) are rows of a compact block of 16-row image stripes stripeFirst, stripeFirst +
// stripeStride, ... (one GPU's share of a frame dealt out in round-robin stripes): row r of the block is
// image row 16 (stripeFirst + (r / 16) stripeStride) + r % 16, accum / rgba are indexed by block row, and
// the random stream is keyed by the image pixel as ever, so the block holds the bytes of those image rows.
template <bool COUNT, int WAVES, bool STRIPES>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void path_trace_persistent_kernel(
    const accel::View view, const kk_pt_scene p, const kk_pt_frame f, float *__restrict__ accum,
    uint8_t *__restrict__ rgba, int W, int H, int y0, int y1, float sampleIdx, int sampleCount, uint32_t seed,
    uint32_t *pixelQueue, unsigned long long *counters, int stripeFirst, int stripeStride) {
    using namespace persist;
    using namespace accel;
    __shared__ uint32_t s_stack[LANE_LDS_DEPTH * THREADS];
    __shared__ unsigned short s_bits[LANE_LDS_DEPTH * THREADS];
    __shared__ unsigned long long s_key[THREADS]; // pooled triangle test: an owner's best (t, sorted position) ...
    __shared__ uint32_t s_mark[THREADS];          // ... which lane's triangles start at a pair ...
    __shared__ uint32_t s_rec[THREADS];           // ... and the record of an owner's winning triangle
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const LaneStack st{s_stack + tid, s_bits + tid, THREADS};
    uint2 deep[LANE_DEPTH - LANE_LDS_DEPTH];

    const kk_lbvh_node *nodes = view.hdr->nodes;
    const char *wideBytes = reinterpret_cast<const char *>(view.wide);
    AABB rootBox;
    rootBox.minPos = {view.hdr->rootBox[0], view.hdr->rootBox[1], view.hdr->rootBox[2]};
    rootBox.maxPos = {view.hdr->rootBox[3], view.hdr->rootBox[4], view.hdr->rootBox[5]};

    const uint32_t tilesX = (static_cast<uint32_t>(W) + 7u) >> 3;
    const uint32_t tilesY = (static_cast<uint32_t>(y1 - y0) + 7u) >> 3;
    // A launch of several samples per pixel asks its queue sixteen times less often than its pixels take
    // to render: one queue serves it (and keeps the kernel's registers for the walk).
    constexpr uint32_t NQ = WAVES == WAVES_LONG ? 1u : PIXEL_QUEUES;
    const uint32_t tileNum = tilesX * tilesY; // queue q hands out the tiles q, q + NQ, ...
    uint32_t queue = NQ == 1u ? 0u : (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu) % NQ, queuesTried = 0; // XCC_ID
    // a lane renders samples sampleFirst .. sampleFirst + sampleCount - 1 of its pixel one after the other
    // (the running mean of a pixel is order dependent) before it takes another pixel
    const uint32_t sampleFirst = static_cast<uint32_t>(sampleIdx);
    const uint32_t sampleEnd = sampleFirst + static_cast<uint32_t>(sampleCount);
    constexpr float RR = .8f;

    uint32_t stage = ST_IDLE;
    bool poolEmpty = false;
    PathState ps{};
    uint32_t outPixel = 0u; // STRIPES: where the lane's pixel sits in the compact block
    RayState ray{};
    RaySigns sg{false, false, false, 0u};
    uint32_t mode = MODE_FAST;
    LaneWalk w{};
    uint32_t pendW = 0u, pendC = 0u;
    Hit8 best{};
    float rayFar = 0.f; // where the lane's ray ends (a shadow ray: just before the light)

    unsigned long long cntPasses = 0, cntWalkLanes = 0, cntShades = 0, cntShadeLanes = 0, cntRays = 0,
                       cntVisits = 0, cntTris = 0, cntRounds = 0, cntPairs = 0, cntExact = 0;
    // a new ray for this lane: `closest` walks for the nearest hit, else for any hit nearer than tMax
    auto start_ray = [&](f3 pos, f3 dir, bool closest, float tMax) {
        if (COUNT)
            ++cntRays;
        best.t = rayFar = closest ? FloatMax : tMax;
        best.bx = best.by = 0.f;
        best.fi = best.k = best.w = 0xFFFFFFFFu;
        best.sure = false;
        pendC = 0u;
        float tEnter, tExit;
        HitAABB(Ray{pos, dir}, rootBox, tEnter, tExit);
        prepare_ray(pos, dir, ray, sg, mode);
        w.start(sg.oct);
        if (tEnter > tExit || !ray_is_finite(pos, dir))
            stage = closest ? ST_DONE_C : ST_DONE_S_CLEAR;
        else
            stage = closest ? ST_WALK_C : ST_WALK_S;
    };

    // What a finished walk means for its path; run at the head of the sections that follow walks.
    auto consume_walk = [&]() {
        // a closest-hit walk is over (ray_tracer_render.cu:321-347)
        if (stage == ST_DONE_C) {
            if (best.t == FloatMax)
                stage = ST_END;
            else {
                const kk_pt_triangle triangle = p.triangles[best.fi];
                ps.grp = triangle.grp_idx;
                const kk_material &matr = p.materials[ps.grp];
                if (matr.emissive) {
                    ps.Lo = ps.Lo + ps.throughout * V(matr.emission);
                    stage = ST_END;
                } else {
                    ps.norm = normalize(best.bx * trav::load_pos(p.normals, triangle.norm_idx[0]) +
                                        best.by * trav::load_pos(p.normals, triangle.norm_idx[1]) +
                                        (1.f - best.bx - best.by) *
                                            trav::load_pos(p.normals, triangle.norm_idx[2]));
                    ps.dirOut = -ray.dir;
                    ray.pos = ray.pos + best.t * ray.dir + Eps * ps.norm;
                    ps.Li = F3(0.f);
                    ps.li = 0u;
                    stage = ST_LIGHTS;
                }
            }
        }
        // a shadow walk is over (:381-389)
        if (stage == ST_DONE_S_CLEAR) {
            const kk_material matr = p.materials[ps.grp];
            const kk_light &lht = p.lights[ps.li];
            ps.Li = ps.Li + ps.fOverPdf * V(lht.radiance) * BRDF(matr, ray.dir, ps.dirOut, ps.norm);
        }
        if (stage == ST_DONE_S_HIT || stage == ST_DONE_S_CLEAR) {
            ++ps.li;
            stage = ST_LIGHTS;
        }

    };
    long long tcLoop = 0, tcV = 0, tcT = 0, tcLights = 0, tcBounce = 0, tcEnd = 0, tc0 = 0, tc1 = 0;
    auto lap = [&](long long &acc) {
        if (COUNT) {
            tc1 = clock64();
            acc += tc1 - tc0;
            tc0 = tc1;
        }
    };
    if (COUNT)
        tc0 = clock64();
    while (true) {
        // ================= walk: one step per lane and pass until no lane has a ray ====================
        while (true) {
            const int nWalk = __popcll(__ballot(stage == ST_WALK_C || stage == ST_WALK_S));
            if (nWalk == 0)
                break;
            if (COUNT) {
                ++cntPasses;
                cntWalkLanes += nWalk;
            }
            lap(tcLoop);
            if (stage == ST_WALK_C || stage == ST_WALK_S) {
                if (pendC == 0u) {
                    uint32_t cur;
                    if (!w.next(st, deep, sg.oct, cur))
                        stage = stage == ST_WALK_S ? ST_DONE_S_CLEAR : ST_DONE_C;
                    else {
                        const float limit = best.t + trav::CULL_SLACK * best.t;
                        bool unusedFlag = false;
                        const Visit vis = visit_record<false>(wideBytes, cur, ray, sg, limit, unusedFlag);
                        pendW = vis.leafW;
                        pendC = vis.leafC;
                        w.enter(st, deep, vis, sg.oct);
                        if (COUNT) {
                            ++cntVisits;
                            if (!(vis.flags & (LONG_RUN | HUGE_RUN))) {
                                for (int sl = 0; sl < 8; ++sl)
                                    if ((vis.leafHits >> sl) & 1u)
                                        cntExact += (vis.counts >> (4 * sl)) & 0xfu;
                            } else
                                cntExact += vis.leafC;
                        }
                    }
                }
                lap(tcV);
            }
            // The triangles the lanes' last visits left pending are tested as one pool, 64 (ray,
            // triangle) pairs at a time: pair i belongs to the first lane whose running total of
            // pending triangles exceeds i, is tested by lane i with that lane's ray, and the owner ends
            // up with what its own one-by-one loop would keep - the smallest (t, sorted position) among
            // its triangles that are hit, nearer than its best and accepted. One long leaf or a few
            // lanes with work do not hold the other lanes up for a pass per triangle.
            while (__ballot((stage == ST_WALK_C || stage == ST_WALK_S) && pendC != 0u) != 0ull) {
                uint32_t tested = 0u, pairs = 0u;
                const bool won = pooled_round<COUNT>(view.recs, nodes, view.parents, s_key + (tid & ~63),
                                                     s_rec + (tid & ~63), s_mark + (tid & ~63), lane, ray, mode,
                                                     stage == ST_WALK_C, stage == ST_WALK_C || stage == ST_WALK_S,
                                                     pendW, pendC, best.t, best.k, best.w, tested, pairs);
                if (won && stage == ST_WALK_S)
                    stage = ST_DONE_S_HIT; // any hit ends a shadow walk
                if (COUNT) {
                    ++cntRounds;
                    cntPairs += pairs;
                    cntTris += tested;
                }
            }
            lap(tcT);
        }
        lap(tcLoop);
        if (__ballot(stage != ST_IDLE) == 0ull && poolEmpty)
            break;

        // ================= shade: every lane takes its path to its next ray =============================
        // A walk in the fast mode took its hits unchecked and kept only (t, sorted position, record) of
        // the best. The hit must pass the test of its leaf's parent box (the triangle's own box lies inside
        // that box and the test is monotonic in the box, so a pass there settles it without loading the
        // parent node), else the lane walks the ray again, checking every hit as it meets it, and consumes
        // nothing this time round. A closest hit that stands gets its distance and barycentrics from the
        // reference's test of that one triangle.
        if ((stage == ST_DONE_C || stage == ST_DONE_S_HIT) && best.w != 0xFFFFFFFFu) {
            const uint4 *tp = reinterpret_cast<const uint4 *>(view.recs + best.w);
            const uint4 a = tp[0], b = tp[1], c = tp[2];
            const f3 v0 = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
            const f3 v1 = F3(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z));
            const f3 v2 = F3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z));
            bool ok = true;
            if (mode == MODE_FAST) {
                AABB tb;
                tb.minPos = {fminf(fminf(v0.x, v1.x), v2.x), fminf(fminf(v0.y, v1.y), v2.y), fminf(fminf(v0.z, v1.z), v2.z)};
                tb.maxPos = {fmaxf(fmaxf(v0.x, v1.x), v2.x), fmaxf(fmaxf(v0.y, v1.y), v2.y), fmaxf(fmaxf(v0.z, v1.z), v2.z)};
                ok = passes_box(tb, ray) || passes_parent(nodes, b.w, ray);
            }
            if (!ok) {
                mode = MODE_CAREFUL;
                best.t = rayFar;
                best.k = best.w = 0xFFFFFFFFu;
                pendC = 0u;
                w.start(sg.oct);
                stage = stage == ST_DONE_C ? ST_WALK_C : ST_WALK_S;
            } else if (stage == ST_DONE_C) {
                best.t = HitTriangle(Ray{ray.pos, ray.dir}, v0, v1, v2, best.bx, best.by);
                best.fi = a.w;
            }
        }
        consume_walk();
        // Each of the three sections below runs when enough lanes wait for it; lanes that are too few
        // wait through the next walk phase (other lanes of the wave still have rays to walk, or the
        // second turn of this loop takes everything that waits).
        int sectionMin = SECTION_MIN;
        {
            const int nL0 = __popcll(__ballot(stage == ST_LIGHTS && ps.li < p.light_num));
            const int nB0 = __popcll(__ballot(stage == ST_LIGHTS && ps.li >= p.light_num));
            const int nE0 = __popcll(__ballot(stage == ST_END || (stage == ST_IDLE && !poolEmpty)));
            const int big = nL0 > nB0 ? (nL0 > nE0 ? nL0 : nE0) : (nB0 > nE0 ? nB0 : nE0);
            if (big < sectionMin)
                sectionMin = big > 0 ? big : 1; // the fullest section always runs
        }
        for (int force = 0; force < 2; ++force) {
        // light samples until the lane has a shadow ray out or is through its lights (:353-380)
        while (true) {
            const int nLights = __popcll(__ballot(stage == ST_LIGHTS && ps.li < p.light_num));
            if (nLights == 0 || (!force && nLights < sectionMin))
                break;
            if (COUNT) {
                ++cntShades;
                cntShadeLanes += nLights;
            }
            if (stage == ST_LIGHTS && ps.li < p.light_num) {
                Rng rng{seed, ps.pixel, ps.sample, ps.draw};
                const kk_light lht = p.lights[ps.li];
                float u0, u1, unused2, unused3;
                rng.next4(u0, u1, unused2, unused3);
                ps.draw = rng.draw;
                LightSample ls = SampleLight(lht, u0, u1, ray.pos);
                ls.pos = ls.pos + Eps * ls.norm;
                const f3 in = ls.pos - ray.pos;
                const float sqrIn = dot(in, in);
                ray.dir = normalize(in);
                const float factor = gmax(dot(ray.dir, ps.norm), 0.f) * gmax(dot(-ray.dir, ls.norm), 0.f) /
                                     sqrIn * lht.area / p.light_tot_area;
                if (factor < FloatEps)
                    ++ps.li;
                else {
                    ps.fOverPdf = factor / ls.pdf;
                    rayFar = sqrtf(sqrIn) - Eps;
                    stage = ST_LAUNCH_S;
                }
            }
        }
        lap(tcLights);
        // after the last light: the path's next direction (:392-425)
        const int nBounce = __popcll(__ballot(stage == ST_LIGHTS && ps.li >= p.light_num));
        if (nBounce != 0 && (force || nBounce >= sectionMin)) {
            if (COUNT) {
                ++cntShades;
                cntShadeLanes += nBounce;
            }
            if (stage == ST_LIGHTS && ps.li >= p.light_num) {
                const kk_material matr = p.materials[ps.grp];
                Rng rng{seed, ps.pixel, ps.sample, ps.draw};
                float pdfIndrct;
                float u, u0, u1, uRoulette;
                rng.next4(u, u0, u1, uRoulette);
                if (u <= matr.kd_over_kd_add_ks) {
                    const f3 tngnt = GenerateTangent(ps.norm);
                    ray.dir = CosineThetaPhiToDirection(sqrtf(1.f - u0), TwoPi * u1, ps.norm, tngnt);
                    pdfIndrct = dot(ray.dir, ps.norm) * OneOverPi * matr.kd_over_kd_add_ks;
                } else {
                    const f3 reflect = normalize(2.f * dot(ps.norm, ps.dirOut) * ps.norm - ps.dirOut);
                    const f3 tngnt = GenerateTangent(reflect);
                    ray.dir = ThetaPhiToDirection(acosf(powf(u0, 1.f / (matr.ns + 1.f))), TwoPi * u1, reflect,
                                                  tngnt);
                    pdfIndrct = (matr.ns + 1.f) * OneOverTwoPi *
                                magnitudePow(gmax(dot(reflect, ray.dir), 0.f), matr.ns) *
                                (1.f - matr.kd_over_kd_add_ks);
                }
                ps.Lo = ps.Lo + ps.throughout * ps.Li;
                ps.throughout = ps.throughout *
                                (dot(ray.dir, ps.norm) / pdfIndrct / RR * BRDF(matr, ray.dir, ps.dirOut, ps.norm));
                ps.draw = rng.draw;
                const bool roulette = uRoulette > RR && ps.depth > 1u;
                ++ps.depth;
                if (!roulette && ps.depth < p.max_path_depth &&
                    gmax(ps.throughout.x, gmax(ps.throughout.y, ps.throughout.z)) >= FloatEps)
                    stage = ST_LAUNCH_C;
                else
                    stage = ST_END;
            }
        }
        lap(tcBounce);
        // finished paths leave, new pixels come in
        const int nWaiting = __popcll(__ballot(stage == ST_END || stage == ST_IDLE));
        if ((__ballot(stage == ST_END) != 0ull || (nWaiting != 0 && !poolEmpty)) &&
            (force || nWaiting >= sectionMin)) {
            if (COUNT) {
                ++cntShades;
                cntShadeLanes += __popcll(__ballot(stage == ST_END || stage == ST_IDLE));
            }
            // tone curve, running mean, RGBA8 (:428-429, 623-637)
            if (stage == ST_END) {
                f3 Lo = ps.Lo / (ps.Lo + 1.f);
                Lo.x = magnitudePow(Lo.x, 1.f / 2.2f);
                Lo.y = magnitudePow(Lo.y, 1.f / 2.2f);
                Lo.z = magnitudePow(Lo.z, 1.f / 2.2f);
                const uint32_t idx = STRIPES ? outPixel : ps.pixel;
                const f3 prev = {accum[3ull * idx], accum[3ull * idx + 1], accum[3ull * idx + 2]};
                const f3 color = prev + (Lo - prev) / static_cast<float>(ps.sample);
                accum[3ull * idx] = color.x;
                accum[3ull * idx + 1] = color.y;
                accum[3ull * idx + 2] = color.z;
                uchar4 o;
                o.x = static_cast<uint8_t>(255.f * color.x);
                o.y = static_cast<uint8_t>(255.f * color.y);
                o.z = static_cast<uint8_t>(255.f * color.z);
                o.w = 255;
                reinterpret_cast<uchar4 *>(rgba)[idx] = o;
                ++ps.sample;
                stage = ps.sample < sampleEnd ? ST_NEXT_SAMPLE : ST_IDLE;
            }
            // idle lanes take the next pixels of the queue (:592-619)
            const unsigned long long idleMask = __ballot(stage == ST_IDLE);
            if (idleMask != 0ull && !poolEmpty) {
                const uint32_t want = __popcll(idleMask);
                const int leader = __ffsll(static_cast<long long>(idleMask)) - 1;
                const uint32_t qLen = (tileNum + NQ - 1u - queue) / NQ * 64u;
                const uint32_t firstTile = queue; // before `queue` moves on
                uint32_t base = 0;
                if (lane == leader)
                    base = atomicAdd(pixelQueue + queue * HEAD_STRIDE, want);
                base = __shfl(base, leader, 64);
                if (base + want >= qLen || base + want < base) { // used up: the next refill asks the next queue
                    queue = queue + 1u == NQ ? 0u : queue + 1u;
                    poolEmpty = ++queuesTried == NQ; // wave-uniform
                }
                if (stage == ST_IDLE) {
                    // idle lanes below this one (mbcnt: no lane mask kept in registers)
                    const uint32_t q = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idleMask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idleMask), 0u));
                    if (q < qLen && q >= base) {
                        const uint32_t tile = (q >> 6) * NQ + firstTile, in = q & 63u;
                        const uint32_t tx = tile % tilesX, ty = tile / tilesX;
                        const int x = static_cast<int>(tx * 8u + (in & 7u));
                        int y = y0 + static_cast<int>(ty * 8u + (in >> 3));
                        bool inside = x < W && y < y1;
                        if (STRIPES) {
                            outPixel = static_cast<uint32_t>(y) * W + x;
                            y = ((stripeFirst + (y >> 4) * stripeStride) << 4) + (y & 15);
                            inside = inside && y < H;
                        }
                        if (inside) {
                            ps.pixel = static_cast<uint32_t>(y) * W + x;
                            ps.sample = sampleFirst;
                            stage = ST_NEXT_SAMPLE;
                        }
                    }
                }
            }
            // the camera ray of a pixel's next sample (:592-619)
            if (stage == ST_NEXT_SAMPLE) {
                const int x = static_cast<int>(ps.pixel % static_cast<uint32_t>(W));
                const int y = static_cast<int>(ps.pixel / static_cast<uint32_t>(W));
                Rng rng{seed, ps.pixel, ps.sample, 0};
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
                float jx, jy, unused2, unused3;
                rng.next4(jx, jy, unused2, unused3);
                tmp.x = 1.f - jx + x;
                tmp.y = 1.f - jy + y;
                tmp.z = static_cast<float>(W);
                tmp.w = static_cast<float>(H);
                tmp.x = (2.f * tmp.x - tmp.z) / tmp.z;
                tmp.y = (2.f * tmp.y - tmp.w) / tmp.w;
                tmp.z = 1.f;
                tmp.w = 1.f;
                tmp = invProj * tmp;
                f3 dir = {tmp.x, tmp.y, tmp.z};
                dir = rot * normalize(dir);
                // the reference transforms a temporary by w2s and discards it (:612-614)
                dir = normalize(dir);
                const f4 pp = w2s * f4{f.eye_pos[0], f.eye_pos[1], f.eye_pos[2], 1.f};
                ps.draw = rng.draw;
                ps.depth = 0u;
                ps.throughout = F3(1.f);
                ps.Lo = F3(0.f);
                // renderScene's loop test before the first bounce (:318-320)
                if (p.max_path_depth > 0) {
                    ray.pos = F3(pp.x, pp.y, pp.z);
                    ray.dir = dir;
                    stage = ST_LAUNCH_C;
                } else
                    stage = ST_END;
            }
        }
        if (__ballot(stage == ST_LAUNCH_C || stage == ST_LAUNCH_S || stage == ST_WALK_C || stage == ST_WALK_S) != 0ull)
            break; // some lane has a ray to walk: the others can wait
        }
        // the rays the sections above asked for are set up together
        if (stage == ST_LAUNCH_C || stage == ST_LAUNCH_S)
            start_ray(ray.pos, ray.dir, stage == ST_LAUNCH_C, rayFar);
        lap(tcEnd);
    }
    if (COUNT) {
        if (lane == 0) {
            atomicAdd(&counters[8], static_cast<unsigned long long>(tcLoop));
            atomicAdd(&counters[9], static_cast<unsigned long long>(tcV));
            atomicAdd(&counters[10], static_cast<unsigned long long>(tcT));
            atomicAdd(&counters[11], static_cast<unsigned long long>(tcLights));
            atomicAdd(&counters[12], static_cast<unsigned long long>(tcBounce));
            atomicAdd(&counters[13], static_cast<unsigned long long>(tcEnd));
        }
        // [0] passes of the walk loop (per wave), [1] lanes walking in them, [2] shade phases,
        // [3] lanes waiting at their start, [4] rays, [5] records visited, [6] triangles tested
        for (int o = 32; o > 0; o >>= 1) {
            cntRays += __shfl_xor(cntRays, o, 64);
            cntVisits += __shfl_xor(cntVisits, o, 64);
            cntTris += __shfl_xor(cntTris, o, 64);
            cntExact += __shfl_xor(cntExact, o, 64);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], cntPasses);
            atomicAdd(&counters[1], cntWalkLanes);
            atomicAdd(&counters[2], cntShades);
            atomicAdd(&counters[3], cntShadeLanes);
            atomicAdd(&counters[4], cntRays);
            atomicAdd(&counters[5], cntVisits);
            atomicAdd(&counters[6], cntTris);
            atomicAdd(&counters[14], cntExact); // triangles of the hit leaf slots alone
            atomicAdd(&counters[7], cntRounds); // rounds of the pooled triangle test (per wave) ...
            atomicAdd(&counters[15], cntPairs); // ... and the pairs tested in them
        }
    }
}

// The variants with WAVES_SHORT waves per SIMD live in an object file of their own: this source compiled with
// KK_PT_SHORT_ONLY under the scheduler that suits them (the Makefile has the measurements), which leaves
// nothing but this launcher in it.
void launch_short_variant(bool count, bool stripes, uint32_t grid, hipStream_t stream, const accel::View &view,
                          const kk_pt_scene &scene, const kk_pt_frame &frame, float *accum, uint8_t *rgba, int W,
                          int H, int y0, int y1, float sampleIdx, int sampleCount, uint32_t seed,
                          uint32_t *pixelQueue, unsigned long long *counters, int stripeFirst, int stripeStride);
#ifdef KK_PT_SHORT_ONLY
void launch_short_variant(bool count, bool stripes, uint32_t grid, hipStream_t stream, const accel::View &view,
                          const kk_pt_scene &scene, const kk_pt_frame &frame, float *accum, uint8_t *rgba, int W,
                          int H, int y0, int y1, float sampleIdx, int sampleCount, uint32_t seed,
                          uint32_t *pixelQueue, unsigned long long *counters, int stripeFirst, int stripeStride) {
#define KK_SHORT(COUNT, STRIPES)                                                                       \
    path_trace_persistent_kernel<COUNT, persist::WAVES_SHORT, STRIPES><<<grid, THREADS, 0, stream>>>(  \
        view, scene, frame, accum, rgba, W, H, y0, y1, sampleIdx, sampleCount, seed, pixelQueue, counters, \
        stripeFirst, stripeStride)
    if (count) {
        if (stripes)
            KK_SHORT(true, true);
        else
            KK_SHORT(true, false);
    } else {
        if (stripes)
            KK_SHORT(false, true);
        else
            KK_SHORT(false, false);
    }
#undef KK_SHORT
}
} // namespace pt
} // namespace kk
#else
__global__ void rng_uniforms_kernel(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t count,
                                    float *out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Rng rng{seed, pixel, sample, 0};
        for (uint32_t i = 0; i < count; i += 4) {
            float u[4];
            rng.next4(u[0], u[1], u[2], u[3]);
            for (uint32_t j = 0; j < 4 && i + j < count; ++j)
                out[i + j] = u[j];
        }
    }
}

static std::atomic<unsigned long long *> g_debugCounters{nullptr};

// Device address of this GPU's copy of the queue heads (every GPU has its own), asked for once.
static int persist_queue_ptr(uint32_t **out) {
    constexpr int MAX_DEVICES = 64;
    static std::mutex mutex;
    static uint32_t *ptrs[MAX_DEVICES] = {};
    int dev = -1;
    KK_HIP_TRY(hipGetDevice(&dev));
    KK_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index out of range");
    std::lock_guard<std::mutex> lock(mutex);
    if (!ptrs[dev]) {
        void *q = nullptr;
        KK_HIP_TRY(hipGetSymbolAddress(&q, HIP_SYMBOL(persist::g_pixelQueue)));
        KK_REQUIRE(q != nullptr, "no device address for the pixel queue heads");
        ptrs[dev] = static_cast<uint32_t *>(q);
    }
    *out = ptrs[dev];
    return KK_OK;
}

} // namespace pt
} // namespace kk

namespace kk {
namespace pt {

// Rows [y0, y1) of the image (stripeStride == 0), or the 16-row stripes stripeFirst, stripeFirst +
// stripeStride, ... written one after the other into compact arrays (stripeStride > 0).
static int render_samples(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                          const kk_pt_frame *h_frame, float *d_accum, uint8_t *d_rgba, int width, int height,
                          int y0, int y1, int stripeFirst, int stripeStride, float sample_index,
                          int sample_count, uint32_t seed, void *stream_) {
    KK_REQUIRE(d_lbvh && h_scene && h_frame && d_accum && d_rgba, "null pointer");
    KK_REQUIRE(width > 0 && height > 0, "bad image size");
    KK_REQUIRE(sample_index >= 1.f, "sample_index is 1-based");
    KK_REQUIRE(sample_count >= 1 && sample_count <= 65536 && sample_index < 16777216.f - 65536.f,
               "sample_count out of range");
    KK_REQUIRE(h_scene->triangles && h_scene->tri_to_position_indices && h_scene->positions &&
                   h_scene->normals && h_scene->materials,
               "scene arrays missing");
    KK_REQUIRE(h_scene->light_num == 0 || h_scene->lights, "lights missing");
    const bool stripes = stripeStride > 0;
    int stripeNum = 0; // stripes of this launch
    if (stripes) {
        const int all = div_up(height, 16);
        stripeNum = stripeFirst < all ? (all - stripeFirst + stripeStride - 1) / stripeStride : 0;
        y0 = 0;
        y1 = 16 * stripeNum; // rows of the compact block
    }
    if (y0 == y1)
        return KK_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (h_scene->trace_accel) {
        KK_REQUIRE(h_scene->accel_face_num > 0 && h_scene->accel_compacted_face_num > 0 &&
                       h_scene->accel_compacted_face_num <= h_scene->accel_face_num,
                   "trace accel sizes out of range");
        const accel::View view =
            accel::make_view(h_scene->trace_accel, h_scene->accel_face_num, h_scene->accel_compacted_face_num);
        // resident workgroups that pull pixels from a queue; the queue head is one of a ring of words
        // in the library's device image, cleared on this stream before the launch
        uint32_t *queues = nullptr;
        const int rc = persist_queue_ptr(&queues);
        if (rc != KK_OK)
            return rc;
        static std::atomic<uint32_t> nextSlot{0};
        uint32_t *head = queues + persist::SLOT_WORDS * (nextSlot.fetch_add(1u) % persist::COUNTER_SLOTS);
        KK_HIP_TRY(hipMemsetAsync(head, 0, sizeof(uint32_t) * persist::SLOT_WORDS, stream));
        const uint32_t queueLen = div_up(width, 8) * div_up(y1 - y0, 8) * 64u;
        uint32_t pgrid = div_up(queueLen, THREADS);
        const bool longLaunch = sample_count >= persist::LONG_LAUNCH_SAMPLES;
        const uint32_t resident = static_cast<uint32_t>(longLaunch ? persist::WAVES_LONG : persist::WAVES_SHORT) *
                                  static_cast<uint32_t>(device_cu_count());
        if (pgrid > resident)
            pgrid = resident;
        unsigned long long *dbg = g_debugCounters.load();
#define KK_LAUNCH_PT(COUNT, WAVES, STRIPES)                                                           \
    path_trace_persistent_kernel<COUNT, WAVES, STRIPES><<<pgrid, THREADS, 0, stream>>>(                  \
        view, *h_scene, *h_frame, d_accum, d_rgba, width, height, y0, y1, sample_index, sample_count, seed, \
        head, dbg, stripeFirst, stripeStride)
#define KK_LAUNCH_PT2(COUNT, WAVES)                                                                   \
    do {                                                                                               \
        if (stripes)                                                                                   \
            KK_LAUNCH_PT(COUNT, WAVES, true);                                                          \
        else                                                                                           \
            KK_LAUNCH_PT(COUNT, WAVES, false);                                                         \
    } while (0)
        if (!longLaunch)
            launch_short_variant(dbg != nullptr, stripes, pgrid, stream, view, *h_scene, *h_frame, d_accum, d_rgba,
                                 width, height, y0, y1, sample_index, sample_count, seed, head, dbg, stripeFirst,
                                 stripeStride);
        else if (dbg)
            KK_LAUNCH_PT2(true, persist::WAVES_LONG);
        else
            KK_LAUNCH_PT2(false, persist::WAVES_LONG);
#undef KK_LAUNCH_PT2
#undef KK_LAUNCH_PT
    } else
        for (int s = 0; s < sample_count; ++s) { // a launch per sample (and stripe), in order
            const float si = sample_index + static_cast<float>(s);
            if (!stripes) {
                path_trace_kernel<false><<<dim3(div_up(width, TILE_W), div_up(y1 - y0, TILE_H)), THREADS, 0, stream>>>(
                    d_lbvh, accel::View{}, *h_scene, *h_frame, d_accum, d_rgba, width, height, y0, y1, si, seed, 0);
                KK_LAUNCH_CHECK();
            } else
                for (int j = 0; j < stripeNum; ++j) {
                    const int r0 = 16 * (stripeFirst + j * stripeStride);
                    const int r1 = r0 + 16 < height ? r0 + 16 : height;
                    path_trace_kernel<false><<<dim3(div_up(width, TILE_W), 1), THREADS, 0, stream>>>(
                        d_lbvh, accel::View{}, *h_scene, *h_frame, d_accum, d_rgba, width, height, r0, r1, si, seed,
                        r0 - 16 * j);
                    KK_LAUNCH_CHECK();
                }
        }
    KK_LAUNCH_CHECK();
    return KK_OK;
}

} // namespace pt
} // namespace kk

extern "C" int kk_pt_render_samples(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                                    const kk_pt_frame *h_frame, float *d_accum, uint8_t *d_rgba,
                                    int width, int height, int y0, int y1, float sample_index,
                                    int sample_count, uint32_t seed, void *stream_) {
    KK_REQUIRE(width > 0 && height > 0 && y0 >= 0 && y1 <= height && y0 <= y1, "bad image size or rows");
    return kk::pt::render_samples(d_lbvh, h_scene, h_frame, d_accum, d_rgba, width, height, y0, y1, 0, 0,
                                  sample_index, sample_count, seed, stream_);
}

extern "C" int kk_pt_render_stripes_compact(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                                            const kk_pt_frame *h_frame, float *d_accum_stripes,
                                            uint8_t *d_stripes, int width, int height, int stripe_first,
                                            int stripe_stride, float sample_index, int sample_count,
                                            uint32_t seed, void *stream_) {
    KK_REQUIRE(stripe_first >= 0 && stripe_stride > 0, "bad stripe selection");
    return kk::pt::render_samples(d_lbvh, h_scene, h_frame, d_accum_stripes, d_stripes, width, height, 0, 0,
                                  stripe_first, stripe_stride, sample_index, sample_count, seed, stream_);
}

extern "C" int kk_pt_render_sample(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                                   const kk_pt_frame *h_frame, float *d_accum, uint8_t *d_rgba,
                                   int width, int height, int y0, int y1, float sample_index,
                                   uint32_t seed, void *stream_) {
    return kk_pt_render_samples(d_lbvh, h_scene, h_frame, d_accum, d_rgba, width, height, y0, y1,
                                sample_index, 1, seed, stream_);
}

extern "C" int kk_pt_set_work_counters(unsigned long long *d_counters) {
    kk::pt::g_debugCounters.store(d_counters);
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
#endif // KK_PT_SHORT_ONLY
