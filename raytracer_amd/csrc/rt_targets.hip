// The inspection render targets of RayTracer::Render (src/ray_tracer/ray_tracer_render.cu:461-558):
// AABBs (boxes of one tree level, :8-100), Triangles / Positions / Normals / TextureCoords (closest hit
// shaded by barycentrics, position, normal, :508-551) and Lights (:272-310). One thread per pixel,
// 16 x 16 tiles, eye ray through the pixel centre. Closest hits use the walk of traverse.h, which
// returns the hit the reference's left-to-right walk keeps.
#include "kk_host.h"
#include "traverse.h"

namespace kk {
namespace rtt {

constexpr int TILE = 16;
constexpr int THREADS = TILE * TILE;
constexpr int AABB_STACK = 32; // the reference's stack holds 30; depth <= displayTreeHeight + 1

KK_D f3 V(const float *p) { return {p[0], p[1], p[2]}; }

// static_cast<uint8_t>(float) as the reference's device code does it: toward zero into 32 unsigned bits
// with saturation (negative and NaN -> 0), then the low byte
KK_D uint8_t to_u8(float v) {
    const uint32_t u = v > 0.f ? (v < 4294967040.f ? static_cast<uint32_t>(v) : 0xFFFFFFFFu) : 0u;
    return static_cast<uint8_t>(u);
}
KK_D uchar4 to_rgba(f3 c) { return make_uchar4(to_u8(255.f * c.x), to_u8(255.f * c.y), to_u8(255.f * c.z), 255); }

KK_D AABB node_box(const kk_lbvh_node *nodes, uint32_t i) { return trav::load_node(nodes, i).box; }

// ray_tracer_render.cu:8-100
KK_D uchar4 renderAABB(const kk_lbvh_device_data &lbvh, uint8_t displayTreeHeight, const Ray &eyeRay,
                       uint32_t *stkNode, uint8_t *stkCnt, int stride) {
    const f3 rootMin = V(lbvh.root_aabb_min), rootMax = V(lbvh.root_aabb_max);
    f3 extent = rootMax - rootMin;
    for (int a = 0; a < 3; ++a)
        if (extent[a] < FloatEps)
            extent[a] = FloatEps;
    const kk_lbvh_node *nodes = lbvh.internal_nodes;
    float tEnter, tExit;
    HitAABB(eyeRay, node_box(nodes, 0), tEnter, tExit);
    if (tEnter > tExit)
        return make_uchar4(0, 0, 0, 0);
    constexpr float EnterAlpha = .7f;
    constexpr float ExitAlpha = 1.f - EnterAlpha;
    f3 color = F3(-1.f);
    if (displayTreeHeight == 0) {
        const f3 enterPos = eyeRay.pos + tEnter * eyeRay.dir - rootMin;
        const f3 exitPos = eyeRay.pos + tExit * eyeRay.dir - rootMin;
        color = (EnterAlpha * enterPos + ExitAlpha * exitPos) / extent;
        return to_rgba(color);
    }
    int sz = 1;
    stkNode[0] = 0;
    stkCnt[0] = 0;
    float tNearest = FloatMax;
    while (sz != 0) {
        const int top = (sz - 1) * stride;
        if (stkCnt[top] == 2) {
            --sz;
            continue;
        }
        const uint2 ch = *reinterpret_cast<const uint2 *>(nodes + stkNode[top]);
        const uint32_t chIdx = stkCnt[top] == 0 ? ch.x : ch.y;
        ++stkCnt[top];
        if (chIdx & trav::LEAF)
            continue;
        const AABB box = node_box(nodes, chIdx);
        HitAABB(eyeRay, box, tEnter, tExit);
        if (tEnter <= tExit) {
            if (displayTreeHeight == sz - 1 && tEnter < tNearest) {
                f3 ext = box.maxPos - box.minPos;
                for (int a = 0; a < 3; ++a)
                    if (ext[a] < FloatEps)
                        ext[a] = FloatEps;
                const f3 enterPos = eyeRay.pos + tEnter * eyeRay.dir - box.minPos;
                const f3 exitPos = eyeRay.pos + tExit * eyeRay.dir - box.minPos;
                color = (EnterAlpha * enterPos + ExitAlpha * exitPos) / ext;
                tNearest = tEnter;
            } else if (sz <= displayTreeHeight && sz < AABB_STACK) {
                stkNode[sz * stride] = chIdx;
                stkCnt[sz * stride] = 0;
                ++sz;
            }
        }
    }
    if (tNearest == FloatMax)
        return make_uchar4(0, 0, 0, 0);
    return to_rgba(color);
}

// include/cg/ray.h:61-86 with glm::intersectRayPlane (3rd/glm/gtx/intersect.inl:6-26)
KK_D float hitQuad(const Ray &r, f3 o, f3 u, f3 v, float &hu, float &hv) {
    const f3 n = normalize(cross(u, v));
    const float d = dot(r.dir, n);
    float t = -1.f;
    if (fabsf(d) > FloatEps) {
        const float tmp = dot(o - r.pos, n) / d;
        if (tmp > 0.f)
            t = tmp;
    }
    if (t < 0.f)
        return -1.f;
    const f3 xLocal = r.pos + t * r.dir - o;
    const float lenU = length(u), lenV = length(v);
    hu = dot(xLocal, u / lenU);
    hv = dot(xLocal, v / lenV);
    if (hu < 0.f || hu > lenU || hv < 0.f || hv > lenV)
        return -1.f;
    return t;
}
// include/cg/ray.h:88-106
KK_D void hitSphere(const Ray &r, float rSqr, f3 o, float &tEnter, float &tExit) {
    const f3 p2o = o - r.pos;
    const float p2oDotDir = dot(p2o, r.dir);
    if (p2oDotDir < 0.f) {
        tEnter = 0.f, tExit = -1.f;
        return;
    }
    const float qSqr = dot(p2o, p2o) - p2oDotDir * p2oDotDir;
    if (qSqr > rSqr) {
        tEnter = 0.f, tExit = -2.f;
        return;
    }
    const float p = sqrtf(rSqr - qSqr);
    tEnter = gmax(p2oDotDir - p, 0.f);
    tExit = p2oDotDir + 2.f * p;
}

// ray_tracer_render.cu:272-310
KK_D uchar4 renderLights(const kk_pt_scene &p, const Ray &eyeRay) {
    f3 color = F3(0.f);
    float tNearest = FloatMax;
    for (uint32_t li = 0; li < p.light_num; ++li) {
        const kk_light &l = p.lights[li];
        if (l.type == 0) {
            float hu = 0.f, hv = 0.f;
            const float t = hitQuad(eyeRay, V(l.quad.o), V(l.quad.u), V(l.quad.v), hu, hv);
            if (t >= 0.f && t < tNearest) {
                color.x = hu / length(V(l.quad.u));
                color.y = hv / length(V(l.quad.v));
                tNearest = t;
            }
        } else {
            float tEnter, tExit;
            hitSphere(eyeRay, l.sphere.r_sqr, V(l.sphere.o), tEnter, tExit);
            if (tEnter <= tExit && tEnter < tNearest) {
                f3 dir = eyeRay.pos + tEnter * eyeRay.dir - V(l.sphere.o);
                dir = normalize(dir);
                color = vabs(dir);
                tNearest = tEnter;
            }
        }
    }
    if (tNearest == FloatMax)
        return make_uchar4(0, 0, 0, 0);
    return to_rgba(color);
}

__global__ __launch_bounds__(THREADS) void rt_target_kernel(const kk_lbvh_device_data *__restrict__ lbvhPtr,
                                                            const kk_pt_scene p, const kk_pt_frame f,
                                                            int target, uint8_t *__restrict__ rgba, int W,
                                                            int H, int y0, int y1) {
    __shared__ uint32_t s_stack[AABB_STACK * THREADS];
    __shared__ uint8_t s_cnt[AABB_STACK * THREADS];
    const int tid = threadIdx.x;
    const int x = blockIdx.x * TILE + (tid & (TILE - 1));
    const int y = y0 + blockIdx.y * TILE + (tid / TILE);
    if (x >= W || y >= y1)
        return;
    const kk_lbvh_device_data lbvh = *lbvhPtr;
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
        tmp.z = static_cast<float>(W);
        tmp.w = static_cast<float>(H);
        tmp.x = (2.f * x + 1.f - tmp.z) / tmp.z;
        tmp.y = (2.f * y + 1.f - tmp.w) / tmp.w;
        tmp.z = 1.f;
        tmp.w = 1.f;
        tmp = invProj * tmp;
        eyeRay.dir = {tmp.x, tmp.y, tmp.z};
        eyeRay.dir = rot * normalize(eyeRay.dir);
        eyeRay.dir = normalize(eyeRay.dir); // the w2s-transformed temporary is discarded (:487-489)
        const f4 pp = w2s * f4{f.eye_pos[0], f.eye_pos[1], f.eye_pos[2], 1.f};
        eyeRay.pos = {pp.x, pp.y, pp.z};
    }
    uchar4 out;
    if (target == KK_RT_TARGET_AABBS) {
        out = renderAABB(lbvh, p.display_tree_height, eyeRay, s_stack + tid, s_cnt + tid, THREADS);
    } else if (target == KK_RT_TARGET_LIGHTS) {
        out = renderLights(p, eyeRay);
    } else {
        trav::Scene s;
        s.nodes = lbvh.internal_nodes;
        s.faces = lbvh.faces;
        s.compacted = lbvh.compacted_faces;
        s.faceNum = lbvh.face_num;
        s.compactedNum = lbvh.compacted_face_num;
        s.positions = p.positions;
        s.tri = p.tri_to_position_indices;
        uint32_t nf = 0, nt = 0;
        const trav::Hit h = trav::closest_hit<false>(s, eyeRay, s_stack + tid, THREADS, nf, nt);
        if (h.fi == 0xFFFFFFFFu) {
            out = make_uchar4(0, 0, 0, 0);
        } else {
            f3 color = F3(0.f);
            const float bx = h.bx, by = h.by;
            if (target == KK_RT_TARGET_TRIANGLES) {
                color = {bx, by, 1.f - bx - by};
            } else if (target == KK_RT_TARGET_POSITIONS) {
                const f3 pos = eyeRay.pos + h.t * eyeRay.dir;
                const f3 rootMin = V(lbvh.root_aabb_min), rootMax = V(lbvh.root_aabb_max);
                color = (pos - rootMin) / (rootMax - rootMin);
            } else if (target == KK_RT_TARGET_NORMALS) {
                // norms[0] takes both of the first two weights, as in the reference (:537)
                const kk_pt_triangle &face = p.triangles[h.fi];
                const f3 n0 = V(p.normals + 3ull * face.norm_idx[0]), n2 = V(p.normals + 3ull * face.norm_idx[2]);
                color = vabs(bx * n0 + by * n0 + (1.f - bx - by) * n2);
            } else {
                // TextureCoords reads the NORMALS array with the texture-coordinate indices (:547-551)
                const kk_pt_triangle &face = p.triangles[h.fi];
                const float *a = p.normals + 3ull * face.tex_coord_idx[0];
                const float *c = p.normals + 3ull * face.tex_coord_idx[2];
                const float w = 1.f - bx - by;
                color.x = fabsf(bx * a[0] + by * a[0] + w * c[0]);
                color.y = fabsf(bx * a[1] + by * a[1] + w * c[1]);
            }
            out = to_rgba(color);
        }
    }
    reinterpret_cast<uchar4 *>(rgba)[static_cast<size_t>(y) * W + x] = out;
}

} // namespace rtt
} // namespace kk

extern "C" int kk_rt_render_target(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                                   const kk_pt_frame *h_frame, int target, uint8_t *d_rgba, int width,
                                   int height, int y0, int y1, void *stream_) {
    using namespace kk;
    using namespace kk::rtt;
    KK_REQUIRE(d_lbvh && h_scene && h_frame && d_rgba, "null pointer");
    KK_REQUIRE(width > 0 && height > 0 && y0 >= 0 && y1 <= height && y0 <= y1, "bad image size or rows");
    KK_REQUIRE(target >= KK_RT_TARGET_AABBS && target <= KK_RT_TARGET_TEXTURE_COORDS, "unknown render target");
    if (target != KK_RT_TARGET_AABBS && target != KK_RT_TARGET_LIGHTS)
        KK_REQUIRE(h_scene->tri_to_position_indices && h_scene->positions, "scene arrays missing");
    if (target == KK_RT_TARGET_NORMALS || target == KK_RT_TARGET_TEXTURE_COORDS)
        KK_REQUIRE(h_scene->triangles && h_scene->normals, "scene arrays missing");
    KK_REQUIRE(h_scene->light_num == 0 || h_scene->lights, "lights missing");
    if (y0 == y1)
        return KK_OK;
    dim3 grid(div_up(width, TILE), div_up(y1 - y0, TILE));
    rt_target_kernel<<<grid, THREADS, 0, static_cast<hipStream_t>(stream_)>>>(
        d_lbvh, *h_scene, *h_frame, target, d_rgba, width, height, y0, y1);
    KK_LAUNCH_CHECK();
    return KK_OK;
}
