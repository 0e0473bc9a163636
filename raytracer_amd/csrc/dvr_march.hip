// Direct volume rendering of a Depth-Box VDB: RayCaster::RenderDepthBoxVDB<T>, Scene target and the
// three inspection targets (src/ray_caster/ray_caster_dpbx_vdb.cu:44-437, 465-531; VDBStack / HDDA3D / DepthDDA2D of
// include/ray_caster/dda.h), one lane per pixel, a wave covers an 8 x 8 pixel tile so that its lanes
// walk the same bricks.
//
// What replaces the CUDA texture units:
//   * the atlas is stored brick by brick (voxPerAtlasBrick^3 voxels, x fastest, brick k = leaf node k),
//     so the eight voxels of a trilinear sample come from one contiguous 46 KB block; the sample
//     coordinate is still formed in the reference's global atlas coordinates (brickPosInAtlas * 36 + 2
//     + posInBrick, in fp32) and only then reduced to the brick, so it rounds like the reference's;
//   * trilinear filtering in fp32: xB = x - 0.5, i = floor(xB), a = xB - i, value = float(v) / max(T),
//     ((1-a) v0 + a v1) along x, then y, then z (CUDA quantises a to 8 fractional bits; not reproduced);
//   * depth-shell reads are point reads of the element type;
//   * the 256-entry transfer function sits in LDS; lookup x = u * 256 - 0.5, linear, 0 outside.
#include "kk_host.h"
#include "kk_math.h"
#include "radix_hist.h"

extern "C" size_t kk_sort_workspace_bytes(uint32_t n);
extern "C" size_t kk_sort_workspace_error_offset(uint32_t n);

namespace kk {
namespace dvr {

constexpr int THREADS = 256;
constexpr float Eps = .01f; // include/ray_caster/dda.h:11
constexpr uint32_t UndefRltIdx = 0xFFFFFFFFu;

KK_D int sgn(float d) { return d > 0.f ? 1 : d < 0.f ? -1 : 0; }
KK_D f3 toF3(const int p[3]) {
    return {static_cast<float>(p[0]), static_cast<float>(p[1]), static_cast<float>(p[2])};
}
KK_D f3 toF3(i3 p) { return {static_cast<float>(p.x), static_cast<float>(p.y), static_cast<float>(p.z)}; }
KK_D bool isInf(float x) { return fabsf(x) == INFINITY; }

// include/ray_caster/dda.h:35-82
struct HDDA3D {
    i3 sign, mask, chPos;
    float tCurr, tNext;
    f3 tSide, tDlt;

    // dda.h:41-50 divides by voxPerChild and divides voxPerChild by the direction. voxPerChild is a
    // power of two, so scaling by it (or by its exact reciprocal) and by the ray's |1 / dir| rounds as
    // those divisions do.
    KK_D void Prepare(const Ray &ray, f3 absInvDir, f3 min, float voxPerChild) {
        tDlt = absInvDir * voxPerChild;
        // 1 / 2^k, exact: the exponent field mirrored about the bias
        const float invVox = __int_as_float(0x7F000000 - __float_as_int(voxPerChild));
        const f3 pFlt = (ray.pos + tCurr * ray.dir - min) * invVox;
        const f3 fl = vfloor(pFlt);
        tSide = ((fl - pFlt + .5f) * toF3(sign) + .5f) * tDlt + tCurr;
        chPos = {static_cast<int>(fl.x), static_cast<int>(fl.y), static_cast<int>(fl.z)};
    }
    KK_D void Next() {
        mask.x = static_cast<int>((tSide.x < tSide.y) & (tSide.x <= tSide.z));
        mask.y = static_cast<int>((tSide.y < tSide.z) & (tSide.y <= tSide.x));
        mask.z = static_cast<int>((tSide.z < tSide.x) & (tSide.z <= tSide.y));
        tNext = mask.x ? tSide.x : mask.y ? tSide.y : mask.z ? tSide.z : INFINITY;
    }
    // dda.h:74-81 also sets the side distance of an axis with an infinite step to infinity here. It is
    // that already: such an axis has direction 0 and sign 0, so Prepare gave it (0 + 0.5) * inf + tCurr, its
    // mask is never set and nothing is ever added to it. (A denormal direction component - sign +-1, step
    // inf - can give 0 * inf = NaN there; then no axis is ever the nearest, tNext is infinite at once and
    // the walk ends with or without the assignment.)
    KK_D void Step() {
        tCurr = tNext;
        tSide.x = mask.x ? tSide.x + tDlt.x : tSide.x;
        tSide.y = mask.y ? tSide.y + tDlt.y : tSide.y;
        tSide.z = mask.z ? tSide.z + tDlt.z : tSide.z;
        chPos.x += mask.x * sign.x;
        chPos.y += mask.y * sign.y;
        chPos.z += mask.z * sign.z;
    }
};

// include/ray_caster/dda.h:84-172
struct DepthDDA2D {
    i3 sign, mask, posInBrick;
    float t, tStart, dep, tDlt2Dep;
    f3 tSide, tDlt;

    // dda.h:96-150, with the ray's direction signs and |1 / dir| at hand: the face the ray enters
    // through is picked once instead of axis by axis.
    KK_D bool Init(float t_, float maxPosValInBrick, int minDepPos, int maxDepPos, f3 pib, i3 raySign,
                   f3 absInvDir, const Ray &ray) {
        dep = 0.f;
        tDlt2Dep = 0.f;
        sign = raySign;
        mask = {0, 0, 0};
        t = tStart = t_;
        const f3 d = {sign.x == 0 ? INFINITY : sign.x > 0 ? pib.x : maxPosValInBrick - pib.x,
                      sign.y == 0 ? INFINITY : sign.y > 0 ? pib.y : maxPosValInBrick - pib.y,
                      sign.z == 0 ? INFINITY : sign.z > 0 ? pib.z : maxPosValInBrick - pib.z};
        // the face the ray enters through, if it is within half a voxel of it
        const int axis = d.x < d.y && d.x <= d.z ? 0 : d.y < d.z && d.y <= d.x ? 1 : d.z < d.x && d.z <= d.y ? 2 : -1;
        if (axis < 0)
            return false;
        const float dAxis = axis == 0 ? d.x : axis == 1 ? d.y : d.z;
        const int sAxis = axis == 0 ? sign.x : axis == 1 ? sign.y : sign.z;
        if (sAxis == 0 || dAxis >= .5f)
            return false;
        tDlt = absInvDir;
        const f3 fl = vfloor(pib);
        posInBrick = {static_cast<int>(fl.x), static_cast<int>(fl.y), static_cast<int>(fl.z)};
        tSide = ((fl - pib + .5f) * toF3(sign) + .5f) * tDlt + t;
        const int depPos = sAxis == 1 ? minDepPos : maxDepPos;
        if (axis == 0) {
            posInBrick.x = depPos;
            sign.x = 0;
            tSide.x = INFINITY;
            tDlt2Dep = fabsf(ray.dir.x);
        } else if (axis == 1) {
            posInBrick.y = depPos;
            sign.y = 0;
            tSide.y = INFINITY;
            tDlt2Dep = fabsf(ray.dir.y);
        } else {
            posInBrick.z = depPos;
            sign.z = 0;
            tSide.z = INFINITY;
            tDlt2Dep = fabsf(ray.dir.z);
        }
        return true;
    }
    KK_D void StepNext() {
        mask.x = static_cast<int>((tSide.x < tSide.y) & (tSide.x <= tSide.z));
        mask.y = static_cast<int>((tSide.y < tSide.z) & (tSide.y <= tSide.x));
        mask.z = static_cast<int>((tSide.z < tSide.x) & (tSide.z <= tSide.y));
        t = mask.x ? tSide.x : mask.y ? tSide.y : mask.z ? tSide.z : INFINITY;
        dep = tDlt2Dep * (t - tStart);
        // (as in HDDA3D::Step: an axis with an infinite step, or the face's own axis, is at infinity already)
        tSide.x = mask.x ? tSide.x + tDlt.x : tSide.x;
        tSide.y = mask.y ? tSide.y + tDlt.y : tSide.y;
        tSide.z = mask.z ? tSide.z + tDlt.z : tSide.z;
        posInBrick.x += mask.x * sign.x;
        posInBrick.y += mask.y * sign.y;
        posInBrick.z += mask.z * sign.z;
    }
};

template <typename T> struct MaxOf;
template <> struct MaxOf<uint8_t> {
    static constexpr float value = 255.f;
};
template <> struct MaxOf<uint16_t> {
    static constexpr float value = 65535.f;
};

// float(v) / max(T): for 8-bit voxels a 256-entry table in LDS holding exactly these quotients (an
// IEEE division costs about ten instructions, eight of them per sample); 16-bit voxels divide.
template <typename T> struct Norm;
template <> struct Norm<uint8_t> {
    const float *lut; // LDS, lut[v] = float(v) / 255.f
    KK_D float operator()(uint8_t v) const { return lut[v]; }
};
// float(v) / max for a whole number v <= max, max = 255 or 65535, without the division: the product with the
// rounded reciprocal, corrected once by its exact remainder, is the correctly rounded quotient for every
// such v (checked for all 65,536 + 256 of them in exact arithmetic: tests/test_voxel_norm.py) - four
// instructions for the ten of an IEEE division.
template <typename T> KK_D float normDiv(T v) {
    constexpr float m = MaxOf<T>::value, r = 1.f / MaxOf<T>::value;
    const float x = static_cast<float>(v);
    const float q0 = x * r;
    const float e = __builtin_fmaf(-q0, m, x);
    return __builtin_fmaf(e, r, q0);
}
template <> struct Norm<uint16_t> {
    const float *lut; // unused
    KK_D float operator()(uint16_t v) const { return normDiv<uint16_t>(v); }
};

// Trilinear sample of one brick. `sp` is the sample position in global atlas voxel coordinates,
// `origin` the brick's first voxel in those coordinates.
// `originLin` = the brick's first voxel as an offset in a cube of edge A, (z A + y) A + x.
// FLOATLIN: the launch has checked that the offset of any voxel of the atlas in a cube of edge A, and the
// atlas' size in bytes, are below 2^24 and 2^32. The offset of the first tap is then made in floating point
// (whole numbers below 2^24: the two multiply-adds are exact, one conversion instead of three and no
// integer multiplications), and `brick` is the atlas itself with the brick's first voxel in `brickOff`, so
// that the taps are loads from one base pointer for the whole wave plus a 32-bit offset per lane.
#ifndef KK_DVR_FLOAT_OFFSETS
#define KK_DVR_FLOAT_OFFSETS 1 // 0: the integer form everywhere (for comparisons)
#endif
template <typename T> struct Taps {
    T v000, v001, v010, v011, v100, v101, v110, v111; // v<z><y><x>
    float ax, ay, az;                                  // where the sample lies between them
};
template <typename T, bool FLOATLIN = false>
KK_D Taps<T> fetchTaps(const T *__restrict__ brick, int A, int originLin, f3 sp, uint32_t brickOff = 0u) {
    const f3 b = sp - .5f;
    const f3 fl = vfloor(b);
    // The first of the eight taps, as one offset inside the brick (offsets inside a brick fit 32 bits). A
    // ray inside a brick's [0, dims0) box with the apron around it never leaves [0, A-2] along any axis;
    // the clamp of the offset only guards the address against non-finite coordinates.
    const int sy = A, sz = A * A;
    int lin;
    if (FLOATLIN) {
        const float Af = static_cast<float>(A);
        lin = static_cast<int>(__builtin_fmaf(__builtin_fmaf(fl.z, Af, fl.y), Af, fl.x)) - originLin;
    } else
        lin = (static_cast<int>(fl.z) * A + static_cast<int>(fl.y)) * A + static_cast<int>(fl.x) - originLin;
    lin = max(0, min(lin, A * sz - 1 - (sz + sy + 1)));
    const T *p = brick + (static_cast<uint32_t>(lin) + brickOff);
    Taps<T> t;
    t.v000 = p[0], t.v001 = p[1];
    t.v010 = p[sy], t.v011 = p[sy + 1];
    t.v100 = p[sz], t.v101 = p[sz + 1];
    t.v110 = p[sz + sy], t.v111 = p[sz + sy + 1];
    t.ax = b.x - fl.x, t.ay = b.y - fl.y, t.az = b.z - fl.z;
    return t;
}
template <typename T> KK_D float mixTaps(const Taps<T> &t, const Norm<T> &norm) {
    const float v000 = norm(t.v000), v001 = norm(t.v001), v010 = norm(t.v010), v011 = norm(t.v011);
    const float v100 = norm(t.v100), v101 = norm(t.v101), v110 = norm(t.v110), v111 = norm(t.v111);
    const float ax = t.ax, ay = t.ay, az = t.az;
    // The same products and sums as ((1-ax) v0 + ax v1 along x, then y, then z), two at a time: the four
    // mixes along x as two register pairs (one per y), the two along y as one.
    typedef float pair __attribute__((ext_vector_type(2)));
    const pair wx0 = {1.f - ax, 1.f - ax}, wx1 = {ax, ax}, wy0 = {1.f - ay, 1.f - ay}, wy1 = {ay, ay};
    const pair lo = wx0 * pair{v000, v100} + wx1 * pair{v001, v101}; // c00, c10
    const pair hi = wx0 * pair{v010, v110} + wx1 * pair{v011, v111}; // c01, c11
    const pair c = wy0 * lo + wy1 * hi;                               // c0, c1
    const pair z = pair{1.f - az, az} * c;
    return z.x + z.y;
}
template <typename T, bool FLOATLIN = false>
KK_D float sampleBrick(const T *__restrict__ brick, int A, int originLin, f3 sp, const Norm<T> &norm,
                       uint32_t brickOff = 0u) {
    return mixTaps<T>(fetchTaps<T, FLOATLIN>(brick, A, originLin, sp, brickOff), norm);
}

// The table in LDS is padded: entry 0 and entries 257, 258 are zero, entry 1 + i is tf[i], so the two
// taps of a lookup are read without range checks (the border colour of the texture is zero).
constexpr int TF_LDS = 259;
KK_D void tfStage(float4 *s_tf, const float *__restrict__ tf, int tid) { // 256 threads
    s_tf[1 + tid] = reinterpret_cast<const float4 *>(tf)[tid];
    if (tid < 3)
        s_tf[tid == 0 ? 0 : 256 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// The same lookup with the colour as two register pairs (red, green) and (blue, opacity), which is how the
// two 16-byte reads land: the mix of the two taps is then six packed operations and nothing is moved.
typedef float v2f __attribute__((ext_vector_type(2)));
struct TfPairs {
    v2f rg, ba;
};
KK_D TfPairs tfLookupPairs(const float4 *__restrict__ s_tf, float u) {
    const float x = u * 256.f - .5f;
    const float fl = floorf(x);
    const float a = x - fl;
    int i = static_cast<int>(fl);
    i = i < -1 ? -1 : (i > 256 ? 256 : i); // taps outside [0, 255] read zeros
    const float4 t0 = s_tf[i + 1], t1 = s_tf[i + 2];
    const v2f w0 = {1.f - a, 1.f - a}, w1 = {a, a};
    TfPairs r;
    r.rg = v2f{t0.x, t0.y} * w0 + v2f{t1.x, t1.y} * w1;
    r.ba = v2f{t0.z, t0.w} * w0 + v2f{t1.z, t1.w} * w1;
    return r;
}
KK_D f4 tfLookup(const float4 *__restrict__ s_tf, float u) {
    const float x = u * 256.f - .5f;
    const float fl = floorf(x);
    const float a = x - fl;
    int i = static_cast<int>(fl);
    i = i < -1 ? -1 : (i > 256 ? 256 : i); // taps outside [0, 255] read zeros
    const float4 t0 = s_tf[i + 1], t1 = s_tf[i + 2];
    return f4{t0.x, t0.y, t0.z, t0.w} * (1.f - a) + f4{t1.x, t1.y, t1.z, t1.w} * a;
}

// Element `i` of a three-entry kernel-argument array, picked with selects. A run-time index that is the
// same for the whole wave makes the compiler read the element with a scalar load whose byte offset it
// may split over the base and offset operands; the hardware drops the low two bits of each, so the load
// can return a neighbouring element.
template <typename V> KK_D V pick3(const V (&a)[3], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : a[2]; }

struct InLeaf {
    float tCurr, tExit;
};

// AFIX: the atlas brick's edge in voxels when it is known at compile time (36 = 32-voxel leaves with the
// two-voxel apron of a depth-box VDB, the layout of the reference's data sets), 0 = read from the
// parameters. With a constant edge the eight taps of a sample are one address and seven immediate
// offsets.
template <typename T, bool COUNT, bool SHADING, int AFIX>
__global__ __launch_bounds__(THREADS) void dvr_kernel(const kk_vdb_device_data vdb,
                                                      const float *__restrict__ tf,
                                                      const kk_dvr_params rp, const kk_dvr_frame fr,
                                                      uint8_t *__restrict__ rgba, int W, int H, int y0,
                                                      int y1, int stripeFirst, int stripeStride,
                                                      int compact, unsigned long long *counters,
                                                      uint32_t *__restrict__ tileCost,
                                                      const uint32_t *__restrict__ tileOrder) {
    // Which 16 x 16 pixel tile this workgroup renders: its own index, or the entry of `tileOrder` there
    // (a permutation of the launch's tiles, the costly ones first). What a tile took goes to `tileCost`
    // for the order of the next frame. Both optional; the image does not depend on them.
    const long long tileStart = (tileCost != nullptr) ? clock64() : 0ll;
    uint32_t tileId = blockIdx.y * gridDim.x + blockIdx.x;
    if (tileOrder != nullptr) // an entry that is no tile of this launch must not become an address
        tileId = min(tileOrder[tileId], gridDim.x * gridDim.y - 1u);
    const uint32_t tileX = tileId % gridDim.x, tileY = tileId / gridDim.x;
    __shared__ float4 s_tf[TF_LDS];
    __shared__ float s_norm[256];
    __shared__ uint32_t s_tileTime, s_wavesDone;
    const int tid = threadIdx.x;
    tfStage(s_tf, tf, tid);
    s_norm[tid] = normDiv<uint8_t>(static_cast<uint8_t>(tid)); // = float(tid) / 255.f
    if (tid == 0)
        s_tileTime = s_wavesDone = 0u;
    __syncthreads();
    const Norm<T> norm{s_norm};

    // 16 x 16 pixel block = 2 x 2 waves of 8 x 8 pixels
    const int lane = tid & 63, wave = tid >> 6;
    const int x = static_cast<int>(tileX) * 16 + (wave & 1) * 8 + (lane & 7);
    // tile row tileY renders the 16-row stripe number stripeFirst + tileY * stripeStride
    const int y = y0 + (stripeFirst + static_cast<int>(tileY) * stripeStride) * 16 + (wave >> 1) * 8 +
                  (lane >> 3);
    unsigned long long nSamples = 0, nBricks = 0;
    const bool inside = x < W && y < y1;

    if (inside) {
        const kk_vdb_param &p = vdb.vdb_param;
        const kk_vdb_node *__restrict__ nodePools = vdb.node_pools;
        const uint32_t *__restrict__ childPools = vdb.child_pools;
        const T *__restrict__ atlas = static_cast<const T *>(vdb.atlas);
        const int A = AFIX ? AFIX : p.vox_per_atlas_brick;
        constexpr bool FLOAT_OFFSETS = KK_DVR_FLOAT_OFFSETS && AFIX != 0;
        const size_t brickVox = static_cast<size_t>(A) * A * A;
        const bool useDepthBox = p.apron_dep_wid != p.apron_wid;
        const f3 voxPerVol = {p.vox_per_vol[0], p.vox_per_vol[1], p.vox_per_vol[2]};

        // ---- ray generation, ray_caster_dpbx_vdb.cu:475-501 ----
        Ray eyeRay;
        {
            m4 invProj, w2s;
            m3 rot;
            for (int c = 0; c < 4; ++c) {
                invProj.c[c] = {rp.inv_proj[4 * c], rp.inv_proj[4 * c + 1], rp.inv_proj[4 * c + 2],
                                rp.inv_proj[4 * c + 3]};
                w2s.c[c] = {rp.w2s[4 * c], rp.w2s[4 * c + 1], rp.w2s[4 * c + 2], rp.w2s[4 * c + 3]};
            }
            for (int c = 0; c < 3; ++c)
                rot.c[c] = {fr.eye_rot[3 * c], fr.eye_rot[3 * c + 1], fr.eye_rot[3 * c + 2]};
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
            tmp = {eyeRay.dir.x, eyeRay.dir.y, eyeRay.dir.z, 0.f};
            tmp = w2s * tmp;
            eyeRay.dir = normalize(F3(tmp.x, tmp.y, tmp.z));
            tmp = w2s * f4{fr.eye_pos[0], fr.eye_pos[1], fr.eye_pos[2], 1.f};
            eyeRay.pos = {tmp.x, tmp.y, tmp.z};
        }

        f3 rgb = {0.f, 0.f, 0.f};
        float a = 0.f;

        // ---- rayCastVDB, ray_caster_dpbx_vdb.cu:44-137 ----
        bool alive = true;
        float tEnter, tExit;
        uint32_t maxStepCnt = 0;
        float step = 0.f;
        f3 dltPos = {0.f, 0.f, 0.f};
        HitAABB(eyeRay, AABB{F3(0.f), F3(1.f)}, tEnter, tExit);
        alive = !(tEnter >= tExit);
        if (alive) {
            const float tDlt = tExit - tEnter;
            const f3 d = tDlt * eyeRay.dir;
            maxStepCnt = static_cast<uint32_t>(ceilf(
                rp.sampling_rate * length(d) * gmax(voxPerVol.x, gmax(voxPerVol.y, voxPerVol.z))));
            eyeRay.pos = eyeRay.pos * voxPerVol;
            eyeRay.dir = normalize(eyeRay.dir * voxPerVol);
            HitAABB(eyeRay, AABB{F3(0.f), voxPerVol}, tEnter, tExit);
            alive = !(tEnter >= tExit);
        }
        if (alive) {
            const float tDlt = tExit - tEnter;
            step = tDlt / (maxStepCnt - 1);
            dltPos = step * eyeRay.dir;
            HitAABB(eyeRay, AABB{F3(0.f), F3(p.vox_per_vdb)}, tEnter, tExit);
            alive = !(tEnter >= tExit);
        }
        if (alive) {
            // VDBStack, dda.h:13-33. With at most three levels a ray is either in the root's grid or in
            // one level-1 node's grid, so the stack is the state of the grid being walked (kept in
            // registers: extent, shift, child-pool row, exit distance) plus the root's, which every lane
            // shares. `lev` above the root's means the walk is over.
            const int rootLev = p.root_lev;
            const int DONE = rootLev + 1;
            // the per-level parameters are picked with selects: indexing kernel arguments with a
            // run-time level makes the compiler split the byte offset over a scalar load's operands
            const bool three = rootLev == 2;
            const kk_vdb_node root = nodePools[three ? p.node_starts[2] : p.node_starts[1]];
            const f3 rootMin = toF3(root.pos);
            const uint32_t *rootChildren = childPools + (three ? p.child_starts[1] : p.child_starts[0]) +
                                           root.child_pool_offs_relative;
            const int rootDim = three ? p.dims[2] : p.dims[1];
            const int rootL2 = three ? p.log2_dims[2] : p.log2_dims[1];
            const float rootVox = three ? p.voxs_per_child[2] : p.voxs_per_child[1];
            const float tExitRoot = tExit - Eps;
            // grids below the root exist only in a three-level tree
            const int subDim = p.dims[1], subL2 = p.log2_dims[1];
            const float subVox = p.voxs_per_child[1];
            const f3 absInvDir = vabs(1.f / eyeRay.dir);

            int lev = rootLev;
            int dimLev = rootDim, l2 = rootL2;
            const uint32_t *children = rootChildren;
            float tExitLev = tExitRoot;

            HDDA3D h;
            h.sign = {sgn(eyeRay.dir.x), sgn(eyeRay.dir.y), sgn(eyeRay.dir.z)};
            h.mask = {0, 0, 0};
            h.tCurr = tEnter + Eps;
            h.tNext = 0.f;
            h.Prepare(eyeRay, absInvDir, rootMin, rootVox);

            const float dims0f = p.voxs_per_child[1];
            // Pop, dda.h:28-32: leave every grid whose exit lies behind tCurr
            auto ascend = [&]() {
                if (h.tCurr >= tExitLev) {
                    if (lev == rootLev)
                        lev = DONE;
                    else {
                        lev = rootLev;
                        dimLev = rootDim;
                        l2 = rootL2;
                        children = rootChildren;
                        tExitLev = tExitRoot;
                        h.Prepare(eyeRay, absInvDir, rootMin, rootVox);
                        if (h.tCurr >= tExitRoot)
                            lev = DONE;
                    }
                }
            };

            // A wave's lanes do the same kind of work together: every lane runs over empty cells until
            // something happens to it (a child, the grid's end, the node's exit), the lanes deal with
            // that, and only when each of them stands in a leaf or has left the tree do the lanes in
            // leaves skip and sample. `budget` bounds a walk fed with non-finite input.
            uint32_t leafIdx = UndefRltIdx;
            int budget = 1 << 20;
            while (lev != DONE) {
                while (lev != DONE && leafIdx == UndefRltIdx) {
                    // The child of the cell after this one is fetched while this cell's is awaited: its
                    // position follows from the step the DDA would take. `cell` = index in the grid's
                    // child block, or none outside the grid.
                    auto cellOf = [&](i3 c) -> uint32_t {
                        const bool in = static_cast<uint32_t>(c.x) < static_cast<uint32_t>(dimLev) &&
                                        static_cast<uint32_t>(c.y) < static_cast<uint32_t>(dimLev) &&
                                        static_cast<uint32_t>(c.z) < static_cast<uint32_t>(dimLev);
                        const uint32_t rel = (static_cast<uint32_t>(c.z) << (l2 << 1)) |
                                             (static_cast<uint32_t>(c.y) << l2) | static_cast<uint32_t>(c.x);
                        return in ? rel : UndefRltIdx;
                    };
                    uint32_t cell = cellOf(h.chPos);
                    bool inGrid = cell != UndefRltIdx;
                    uint32_t ch = children[inGrid ? cell : 0u];
                    // One cell per pass; the run over empty cells ends at a child, at the grid's end or at
                    // the node's exit. The loop has one way out: `walking` says whether to go on.
                    bool walking = inGrid;
                    while (walking) {
                        h.Next();
                        const uint32_t nextCell = cellOf(i3{h.chPos.x + h.mask.x * h.sign.x,
                                                            h.chPos.y + h.mask.y * h.sign.y,
                                                            h.chPos.z + h.mask.z * h.sign.z});
                        const uint32_t chNext = children[nextCell != UndefRltIdx ? nextCell : 0u];
                        walking = ch == UndefRltIdx;
                        if (walking) {
                            h.Step();
                            // the node's exit comes before the grid's end: the cell stays the one it was then
                            const bool out = h.tCurr >= tExitLev || --budget < 0;
                            if (!out) {
                                inGrid = nextCell != UndefRltIdx;
                                ch = chNext;
                            }
                            walking = !out & inGrid;
                        }
                    }
                    if (!inGrid || budget < 0)
                        lev = DONE;
                    else if (ch == UndefRltIdx)
                        ascend();
                    else if (lev == 1)
                        leafIdx = ch;
                    else {
                        // Push, dda.h:23-27
                        const kk_vdb_node sub = nodePools[p.node_starts[1] + ch];
                        lev = 1;
                        dimLev = subDim;
                        l2 = subL2;
                        children = childPools + p.child_starts[0] + sub.child_pool_offs_relative;
                        tExitLev = h.tNext - Eps;
                        h.tCurr += Eps;
                        h.Prepare(eyeRay, absInvDir, toF3(sub.pos), subVox);
                        ascend();
                    }
                }
                if (lev == DONE)
                    break;

                {
                    const uint32_t ch = leafIdx;
                    leafIdx = UndefRltIdx;
                    if (--budget < 0)
                        break;
                    h.tCurr += Eps;
                    InLeaf param{h.tCurr, h.tNext - Eps};
                    const kk_vdb_node leaf = nodePools[p.node_starts[0] + ch];
                    if (COUNT)
                        ++nBricks;
                    // ---- inLeafCallback, ray_caster_dpbx_vdb.cu:194-273 ----
                    bool keepGoing = true;
                    {
                        param.tCurr = step * ceilf(param.tCurr / step);
                        const f3 brickMinVoxPos = toF3(leaf.pos);
                        f3 posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - brickMinVoxPos;
                        const i3 origin = {leaf.brick_pos_in_atlas[0] * A, leaf.brick_pos_in_atlas[1] * A,
                                           leaf.brick_pos_in_atlas[2] * A};
                        const f3 atlasBrickMinVoxPos = {static_cast<float>(origin.x + p.apron_dep_wid),
                                                        static_cast<float>(origin.y + p.apron_dep_wid),
                                                        static_cast<float>(origin.z + p.apron_dep_wid)};
                        const T *__restrict__ brick = atlas + brickVox * ch;
                        const int originLin = (origin.z * A + origin.y) * A + origin.x;
                        // the samples' taps: from the atlas plus a 32-bit offset where the launch allows it
                        const T *__restrict__ sampleBase = FLOAT_OFFSETS ? atlas : brick;
                        const uint32_t sampleOff = FLOAT_OFFSETS ? static_cast<uint32_t>(brickVox) * ch : 0u;
                        bool skipped = false;
                        if (useDepthBox) {
                            // depthSkip, :139-162
                            DepthDDA2D d;
                            if (d.Init(param.tCurr, dims0f, p.min_dep_pos_val_in_brick,
                                       p.max_dep_pos_val_in_brick, posInBrick, h.sign, absInvDir, eyeRay)) {
                                while (true) {
                                    const int lx = d.posInBrick.x + p.apron_dep_wid;
                                    const int ly = d.posInBrick.y + p.apron_dep_wid;
                                    const int lz = d.posInBrick.z + p.apron_dep_wid;
                                    T dep = 0; // outside the brick: border
                                    if (lx >= 0 && lx < A && ly >= 0 && ly < A && lz >= 0 && lz < A)
                                        dep = brick[(static_cast<uint32_t>(lz) * static_cast<uint32_t>(A) +
                                                     static_cast<uint32_t>(ly)) *
                                                        static_cast<uint32_t>(A) +
                                                    static_cast<uint32_t>(lx)];
                                    if (static_cast<float>(dep) <= d.dep)
                                        break;
                                    if (d.t >= param.tExit) {
                                        skipped = true;
                                        break;
                                    }
                                    param.tCurr = d.t;
                                    d.StepNext();
                                }
                            }
                            if (!skipped) {
                                param.tCurr = step * ceilf(param.tCurr / step);
                                posInBrick = eyeRay.pos + param.tCurr * eyeRay.dir - brickMinVoxPos;
                            }
                        }
                        if (!skipped) {
                            // One sample: value, colour, compositing, and the step to the next one (taken whether
                            // or not the ray ends here: a ray that ends has no use for its position any more,
                            // and the others need no select).
                            auto takeSample = [&]() {
                                f3 samplePos = atlasBrickMinVoxPos + posInBrick;
                                const float scalar = sampleBrick<T, FLOAT_OFFSETS>(sampleBase, A, originLin, samplePos, norm, sampleOff);
                                const TfPairs tfPairs = tfLookupPairs(s_tf, scalar);
                                const f4 tfCol = {tfPairs.rg.x, tfPairs.rg.y, tfPairs.ba.x, tfPairs.ba.y};
                                f3 tfColRGB = {tfCol.x, tfCol.y, tfCol.z};
                                if (COUNT)
                                    ++nSamples;
                                if (SHADING) {
                                    f3 N;
                                    samplePos.x += .5f;
                                    float v0 = sampleBrick<T, FLOAT_OFFSETS>(sampleBase, A, originLin, samplePos, norm, sampleOff);
                                    samplePos.x -= 1.f;
                                    float v1 = sampleBrick<T, FLOAT_OFFSETS>(sampleBase, A, originLin, samplePos, norm, sampleOff);
                                    N.x = v1 - v0;
                                    samplePos.y += .5f;
                                    v0 = sampleBrick<T, FLOAT_OFFSETS>(sampleBase, A, originLin, samplePos, norm, sampleOff);
                                    samplePos.y -= 1.f;
                                    v1 = sampleBrick<T, FLOAT_OFFSETS>(sampleBase, A, originLin, samplePos, norm, sampleOff);
                                    N.y = v1 - v0;
                                    samplePos.z += .5f;
                                    v0 = sampleBrick<T, FLOAT_OFFSETS>(sampleBase, A, originLin, samplePos, norm, sampleOff);
                                    samplePos.z -= 1.f;
                                    v1 = sampleBrick<T, FLOAT_OFFSETS>(sampleBase, A, originLin, samplePos, norm, sampleOff);
                                    N.z = v1 - v0;
                                    N = normalize(N);
                                    if (dot(eyeRay.dir, N) > 0.f)
                                        N = -N;
                                    const f3 lightPos = {rp.light_pos[0], rp.light_pos[1], rp.light_pos[2]};
                                    const f3 lightCol = {rp.light_col[0], rp.light_col[1], rp.light_col[2]};
                                    const f3 p2lDir = normalize(lightPos - brickMinVoxPos - posInBrick);
                                    const f3 ambient = rp.ka * tfColRGB;
                                    const f3 diffuse =
                                        rp.kd * gmax(0.f, dot(N, p2lDir)) * tfColRGB * lightCol;
                                    const f3 hfDir = normalize(p2lDir - eyeRay.dir);
                                    const f3 specular =
                                        rp.ks * powf(gmax(0.f, dot(N, hfDir)), rp.shiness) * lightCol;
                                    tfColRGB = {ambient.x + diffuse.x + specular.x,
                                                ambient.y + diffuse.y + specular.y,
                                                ambient.z + diffuse.z + specular.z};
                                }
                                // rgb += ((1 - a) tfCol.w) tfColRGB, a += (1 - a) tfCol.w: red and green as a pair
                                const float wgt = (1.f - a) * tfCol.w;
                                const v2f rg = v2f{rgb.x, rgb.y} + v2f{wgt, wgt} * v2f{tfColRGB.x, tfColRGB.y};
                                rgb = {rg.x, rg.y, rgb.z + wgt * tfColRGB.z};
                                a = a + wgt;
                                param.tCurr += step;
                                posInBrick = posInBrick + dltPos;
                            };
                            // One test per pass: the conjunction is evaluated whole, not term by term, and the six
                            // bounds as two: no coordinate below 0 = the smallest is not, none at or above the edge
                            // = the largest is not (a NaN coordinate fails no comparison either way: the smallest
                            // and the largest of three skip it, and are NaN themselves when all three are).
                            auto inBrick = [&]() {
                                return (param.tCurr < param.tExit) &
                                       !(__builtin_fminf(__builtin_fminf(posInBrick.x, posInBrick.y), posInBrick.z) < 0.f) &
                                       !(__builtin_fmaxf(__builtin_fmaxf(posInBrick.x, posInBrick.y), posInBrick.z) >= dims0f);
                            };
                            // The reference ends a ray after the sample that takes its opacity to the limit. A ray
                            // that gets here is below a positive limit, so testing before every sample is the same
                            // loop with one way out and no flag to carry; a limit of zero or less (one sample per
                            // ray) keeps the reference's form.
                            if (!SHADING && !COUNT && rp.max_alpha > 0.f) {
                                // The same loop in three stages: the voxels of the sample after next are asked
                                // for, the next sample is mixed and its colour looked up, and the present one is
                                // composited - only the compositing and the opacity test depend on the sample
                                // before, so the loads' way through the caches and the table lookups in LDS run
                                // beside the arithmetic of other samples. Two samples per pass, each stage's
                                // values in variables of their own, so nothing is copied. A ray that ends has
                                // fetched two samples and looked up one that it does not use (offsets are clamped
                                // into the brick whatever the position); its position is not used after the loop.
                                auto fetch = [&]() {
                                    return fetchTaps<T, FLOAT_OFFSETS>(sampleBase, A, originLin,
                                                                       atlasBrickMinVoxPos + posInBrick, sampleOff);
                                };
                                auto lookup = [&](const Taps<T> &t) { return tfLookupPairs(s_tf, mixTaps<T>(t, norm)); };
                                auto blend = [&](const TfPairs &tfc) {
                                    const float wgt = (1.f - a) * tfc.ba.y;
                                    const v2f rg = v2f{rgb.x, rgb.y} + v2f{wgt, wgt} * tfc.rg;
                                    rgb = {rg.x, rg.y, rgb.z + wgt * tfc.ba.x};
                                    a = a + wgt;
                                };
                                auto stepOn = [&]() {
                                    param.tCurr += step;
                                    posInBrick = posInBrick + dltPos;
                                };
                                if (inBrick() & !(a >= rp.max_alpha)) {
                                    Taps<T> t0 = fetch();
                                    stepOn();
                                    bool v1 = inBrick(), v2;
                                    Taps<T> t1 = fetch();
                                    TfPairs c0 = lookup(t0), c1;
                                    while (true) {
                                        stepOn();
                                        v2 = inBrick();
                                        t0 = fetch();
                                        c1 = lookup(t1);
                                        blend(c0);
                                        if (!(v1 & !(a >= rp.max_alpha)))
                                            break;
                                        stepOn();
                                        v1 = inBrick();
                                        t1 = fetch();
                                        c0 = lookup(t0);
                                        blend(c1);
                                        if (!(v2 & !(a >= rp.max_alpha)))
                                            break;
                                    }
                                }
                                asm volatile("" : "+v"(a));
                                keepGoing = !(a >= rp.max_alpha);
                            } else if (rp.max_alpha > 0.f) {
                                while (inBrick() & !(a >= rp.max_alpha))
                                    takeSample();
                                // (asked again of a value the compiler cannot trace back into the loop: it would
                                // otherwise carry the loop's own comparison out of it, three mask operations a sample)
                                asm volatile("" : "+v"(a));
                                keepGoing = !(a >= rp.max_alpha);
                            } else
                                while (inBrick() & keepGoing) {
                                    takeSample();
                                    keepGoing = !(a >= rp.max_alpha);
                                }
                        }
                    }
                    if (!keepGoing)
                        break;
                    h.Next(); // (the masks of the walk's last Next, made again here instead of kept across the leaf)
                    h.Step();
                    ascend();
                }
            }
        }

        // HDRToLDRCorrect + GammaCorrect (include/cg/math.h:75-93), RGBA8 with alpha = opacity
        // a pixel that gathered no opacity is black: the curve maps 0 to 0
        uchar4 o = make_uchar4(0, 0, 0, 0);
        if (a > 0.f) {
            rgb = rgb / (rgb + 1.f);
            rgb.x = powf(rgb.x, 1.f / 2.2f);
            rgb.y = powf(rgb.y, 1.f / 2.2f);
            rgb.z = powf(rgb.z, 1.f / 2.2f);
            o.x = static_cast<uint8_t>(255.f * rgb.x);
            o.y = static_cast<uint8_t>(255.f * rgb.y);
            o.z = static_cast<uint8_t>(255.f * rgb.z);
            o.w = static_cast<uint8_t>(255.f * a);
        }
        // compact: this launch's stripes one after the other (the block a rank sends to the gatherer)
        const int outRow = compact ? static_cast<int>(tileY) * 16 + (wave >> 1) * 8 + (lane >> 3) : y;
        reinterpret_cast<uchar4 *>(rgba)[static_cast<size_t>(outRow) * W + x] = o;
    }

    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1) {
            nSamples += __shfl_xor(nSamples, o, 64);
            nBricks += __shfl_xor(nBricks, o, 64);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], nSamples);
            atomicAdd(&counters[1], nBricks);
        }
    }
    // What the tile took = what its slowest wave took, in units of 64 cycles (so that any frame's tiles fit
    // 16 bits): every wave leaves its time in LDS, and the one that leaves last writes the largest. (No
    // barrier: a wave that is done frees its registers for the next workgroup.)
    if (tileCost != nullptr && lane == 0) {
        const long long took = (clock64() - tileStart) >> 6;
        atomicMax(&s_tileTime, took > 65535ll ? 65535u : static_cast<uint32_t>(took));
        if (atomicAdd(&s_wavesDone, 1u) == THREADS / 64 - 1) // LDS operations of a wave complete in order
            tileCost[tileId] = atomicMax(&s_tileTime, 0u);
    }
}

// ---- RayCaster::RenderRAWVolume (src/ray_caster/ray_caster_raw_volume.cu:27-246) --------------------
// The un-accelerated marcher over the dense volume: every sample is a trilinear read of the raw
// volume (voxel outside the volume = 0, as the border address mode gives), same transfer function,
// shading and compositing as above. Target AABB draws the entry / exit points of the unit box.

template <typename T>
KK_D float sampleVolume(const T *__restrict__ vol, int dx, int dy, int dz, f3 sp) {
    const f3 b = sp - .5f;
    const f3 fl = vfloor(b);
    const float ax = b.x - fl.x, ay = b.y - fl.y, az = b.z - fl.z;
    const int i = static_cast<int>(fl.x), j = static_cast<int>(fl.y), k = static_cast<int>(fl.z);
    float v[2][2][2];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int x = i + (c & 1), y = j + ((c >> 1) & 1), z = k + (c >> 2);
        T t = 0;
        if (x >= 0 && y >= 0 && z >= 0 && x < dx && y < dy && z < dz)
            t = vol[(static_cast<size_t>(z) * dy + y) * dx + x];
        v[c >> 2][(c >> 1) & 1][c & 1] = normDiv<T>(t);
    }
    const float c00 = (1.f - ax) * v[0][0][0] + ax * v[0][0][1];
    const float c01 = (1.f - ax) * v[0][1][0] + ax * v[0][1][1];
    const float c10 = (1.f - ax) * v[1][0][0] + ax * v[1][0][1];
    const float c11 = (1.f - ax) * v[1][1][0] + ax * v[1][1][1];
    const float c0 = (1.f - ay) * c00 + ay * c01;
    const float c1 = (1.f - ay) * c10 + ay * c11;
    return (1.f - az) * c0 + az * c1;
}

template <typename T, bool COUNT>
__global__ __launch_bounds__(THREADS) void raw_kernel(const T *__restrict__ vol, int dx, int dy, int dz,
                                                      const float *__restrict__ tf,
                                                      const kk_dvr_params rp, const kk_dvr_frame fr,
                                                      int target, uint8_t *__restrict__ rgba, int W,
                                                      int H, int y0, int y1,
                                                      unsigned long long *counters) {
    __shared__ float4 s_tf[TF_LDS];
    const int tid = threadIdx.x;
    tfStage(s_tf, tf, tid);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = y0 + static_cast<int>(blockIdx.y) * 16 + (wave >> 1) * 8 + (lane >> 3);
    unsigned long long nSamples = 0;
    if (x < W && y < y1) {
        Ray eyeRay;
        {
            m4 invProj, w2s;
            m3 rot;
            for (int c = 0; c < 4; ++c) {
                invProj.c[c] = {rp.inv_proj[4 * c], rp.inv_proj[4 * c + 1], rp.inv_proj[4 * c + 2],
                                rp.inv_proj[4 * c + 3]};
                w2s.c[c] = {rp.w2s[4 * c], rp.w2s[4 * c + 1], rp.w2s[4 * c + 2], rp.w2s[4 * c + 3]};
            }
            for (int c = 0; c < 3; ++c)
                rot.c[c] = {fr.eye_rot[3 * c], fr.eye_rot[3 * c + 1], fr.eye_rot[3 * c + 2]};
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
            tmp = {eyeRay.dir.x, eyeRay.dir.y, eyeRay.dir.z, 0.f};
            tmp = w2s * tmp;
            eyeRay.dir = normalize(F3(tmp.x, tmp.y, tmp.z));
            tmp = w2s * f4{fr.eye_pos[0], fr.eye_pos[1], fr.eye_pos[2], 1.f};
            eyeRay.pos = {tmp.x, tmp.y, tmp.z};
        }
        uchar4 o = make_uchar4(0, 0, 0, 0);
        float tEnter, tExit;
        HitAABB(eyeRay, AABB{F3(0.f), F3(1.f)}, tEnter, tExit);
        if (!(tEnter > tExit)) {
            if (target == KK_RAW_TARGET_AABB) {
                // renderAABB, ray_caster_raw_volume.cu:131-145
                const f3 enterPos = eyeRay.pos + tEnter * eyeRay.dir;
                const f3 exitPos = eyeRay.pos + tExit * eyeRay.dir;
                const f3 color = .3f * enterPos + .7f * exitPos;
                o = make_uchar4(static_cast<uint8_t>(255.f * color.x), static_cast<uint8_t>(255.f * color.y),
                                static_cast<uint8_t>(255.f * color.z), 255);
            } else {
                const f3 dim = {static_cast<float>(dx), static_cast<float>(dy), static_cast<float>(dz)};
                uint32_t stepCnt = 0;
                const f3 d0 = (tExit - tEnter) * eyeRay.dir;
                // the reference keeps the rounded-up float in a uint32_t
                const uint32_t maxStepCnt = static_cast<uint32_t>(
                    ceilf(rp.sampling_rate * length(d0) * gmax(dim.x, gmax(dim.y, dim.z))));
                eyeRay.pos = eyeRay.pos * dim;
                eyeRay.dir = normalize(eyeRay.dir * dim);
                HitAABB(eyeRay, AABB{F3(0.f), dim}, tEnter, tExit);
                if (!(tEnter > tExit)) {
                    const float step = (tExit - tEnter) / (maxStepCnt - 1);
                    auto inBox = [&]() {
                        return !(eyeRay.pos.x < 0.f || eyeRay.pos.x >= dim.x || eyeRay.pos.y < 0.f ||
                                 eyeRay.pos.y >= dim.y || eyeRay.pos.z < 0.f || eyeRay.pos.z >= dim.z);
                    };
                    tEnter = step * ceilf(tEnter / step);
                    eyeRay.pos = eyeRay.pos + tEnter * eyeRay.dir;
                    if (!inBox())
                        eyeRay.pos = eyeRay.pos + step * eyeRay.dir;
                    f3 rgb = {0.f, 0.f, 0.f};
                    float a = 0.f;
                    while (stepCnt <= maxStepCnt && inBox()) {
                        const float scalar = sampleVolume<T>(vol, dx, dy, dz, eyeRay.pos);
                        const f4 tfCol = tfLookup(s_tf, scalar);
                        f3 tfColRGB = {tfCol.x, tfCol.y, tfCol.z};
                        if (COUNT)
                            ++nSamples;
                        if (rp.use_shading) {
                            f3 N;
                            f3 sp = eyeRay.pos;
                            sp.x += .5f;
                            float v0 = sampleVolume<T>(vol, dx, dy, dz, sp);
                            sp.x -= 1.f;
                            float v1 = sampleVolume<T>(vol, dx, dy, dz, sp);
                            N.x = v1 - v0;
                            sp.y += .5f;
                            v0 = sampleVolume<T>(vol, dx, dy, dz, sp);
                            sp.y -= 1.f;
                            v1 = sampleVolume<T>(vol, dx, dy, dz, sp);
                            N.y = v1 - v0;
                            sp.z += .5f;
                            v0 = sampleVolume<T>(vol, dx, dy, dz, sp);
                            sp.z -= 1.f;
                            v1 = sampleVolume<T>(vol, dx, dy, dz, sp);
                            N.z = v1 - v0;
                            N = normalize(N);
                            if (dot(eyeRay.dir, N) > 0.f)
                                N = -N;
                            const f3 lightPos = {rp.light_pos[0], rp.light_pos[1], rp.light_pos[2]};
                            const f3 lightCol = {rp.light_col[0], rp.light_col[1], rp.light_col[2]};
                            const f3 p2lDir = normalize(lightPos - eyeRay.pos);
                            const f3 ambient = rp.ka * tfColRGB;
                            const f3 diffuse = rp.kd * gmax(0.f, dot(N, p2lDir)) * tfColRGB * lightCol;
                            const f3 hfDir = normalize(p2lDir - eyeRay.dir);
                            const f3 specular = rp.ks * powf(gmax(0.f, dot(N, hfDir)), rp.shiness) * lightCol;
                            tfColRGB = {ambient.x + diffuse.x + specular.x, ambient.y + diffuse.y + specular.y,
                                        ambient.z + diffuse.z + specular.z};
                        }
                        rgb = rgb + (1.f - a) * tfCol.w * tfColRGB;
                        a = a + (1.f - a) * tfCol.w;
                        if (a >= rp.max_alpha)
                            break;
                        ++stepCnt;
                        eyeRay.pos = eyeRay.pos + step * eyeRay.dir;
                    }
                    rgb = rgb / (rgb + 1.f);
                    rgb.x = powf(rgb.x, 1.f / 2.2f);
                    rgb.y = powf(rgb.y, 1.f / 2.2f);
                    rgb.z = powf(rgb.z, 1.f / 2.2f);
                    o = make_uchar4(static_cast<uint8_t>(255.f * rgb.x), static_cast<uint8_t>(255.f * rgb.y),
                                    static_cast<uint8_t>(255.f * rgb.z), static_cast<uint8_t>(255.f * a));
                }
            }
        }
        reinterpret_cast<uchar4 *>(rgba)[static_cast<size_t>(y) * W + x] = o;
    }
    if (COUNT) {
        for (int o = 32; o > 0; o >>= 1)
            nSamples += __shfl_xor(nSamples, o, 64);
        if (lane == 0)
            atomicAdd(&counters[0], nSamples);
    }
}

// ---- inspection targets of RenderDepthBoxVDB (src/ray_caster/ray_caster_dpbx_vdb.cu:284-437) ---------
// AABBs: every node of level displayLev the ray enters adds its entry point, normalised to the node, at
// half opacity (the scene / volume box tests do not cull here). Depths: the depth value on the entry
// face of the displayLeafLayer-th leaf along the ray (red where the ray does not enter through a
// face). PositionsAfterDepthSkip: where sampling starts in that leaf after the depth skip. Colours are
// converted with the saturating float -> u8 cast of the device.
KK_D uint8_t satU8(float v) {
    const uint32_t u = v > 0.f ? (v < 4294967040.f ? static_cast<uint32_t>(v) : 0xFFFFFFFFu) : 0u;
    return static_cast<uint8_t>(u);
}

KK_D Ray makeEyeRay(const kk_dvr_params &rp, const kk_dvr_frame &fr, int x, int y, int W, int H) {
    m4 invProj, w2s;
    m3 rot;
    for (int c = 0; c < 4; ++c) {
        invProj.c[c] = {rp.inv_proj[4 * c], rp.inv_proj[4 * c + 1], rp.inv_proj[4 * c + 2],
                        rp.inv_proj[4 * c + 3]};
        w2s.c[c] = {rp.w2s[4 * c], rp.w2s[4 * c + 1], rp.w2s[4 * c + 2], rp.w2s[4 * c + 3]};
    }
    for (int c = 0; c < 3; ++c)
        rot.c[c] = {fr.eye_rot[3 * c], fr.eye_rot[3 * c + 1], fr.eye_rot[3 * c + 2]};
    Ray r;
    f4 tmp;
    tmp.z = static_cast<float>(W);
    tmp.w = static_cast<float>(H);
    tmp.x = (2.f * x + 1.f - tmp.z) / tmp.z;
    tmp.y = (2.f * y + 1.f - tmp.w) / tmp.w;
    tmp.z = 1.f;
    tmp.w = 1.f;
    tmp = invProj * tmp;
    r.dir = {tmp.x, tmp.y, tmp.z};
    r.dir = rot * normalize(r.dir);
    tmp = {r.dir.x, r.dir.y, r.dir.z, 0.f};
    tmp = w2s * tmp;
    r.dir = normalize(F3(tmp.x, tmp.y, tmp.z));
    tmp = w2s * f4{fr.eye_pos[0], fr.eye_pos[1], fr.eye_pos[2], 1.f};
    r.pos = {tmp.x, tmp.y, tmp.z};
    return r;
}

template <typename T>
__global__ __launch_bounds__(THREADS) void vdb_target_kernel(const kk_vdb_device_data vdb,
                                                             const kk_dvr_params rp,
                                                             const kk_dvr_frame fr, int target,
                                                             int displayLev, int displayLeafLayer,
                                                             uint8_t *__restrict__ rgba, int W, int H,
                                                             int y0, int y1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = y0 + static_cast<int>(blockIdx.y) * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (x >= W || y >= y1)
        return;
    const kk_vdb_param &p = vdb.vdb_param;
    const T *__restrict__ atlas = static_cast<const T *>(vdb.atlas);
    const int A = p.vox_per_atlas_brick;
    const size_t brickVox = static_cast<size_t>(A) * A * A;
    const f3 voxPerVol = {p.vox_per_vol[0], p.vox_per_vol[1], p.vox_per_vol[2]};
    const float dims0f = p.voxs_per_child[1];
    const bool cull = target != KK_VDB_TARGET_AABBS;
    constexpr float AlphaPerLev = .5f;

    Ray ray = makeEyeRay(rp, fr, x, y, W, H);
    f3 rgb = {0.f, 0.f, 0.f};
    float a = 0.f, step = 0.f;
    int layer = 0;

    // point read of the depth shell at brick-local voxel q (0 outside the brick)
    auto depthAt = [&](const T *brick, i3 q) -> T {
        const int lx = q.x + p.apron_dep_wid, ly = q.y + p.apron_dep_wid, lz = q.z + p.apron_dep_wid;
        if (lx < 0 || lx >= A || ly < 0 || ly >= A || lz < 0 || lz >= A)
            return 0;
        return brick[(static_cast<size_t>(lz) * A + ly) * A + lx];
    };
    // one node of the shown level: its entry point, normalised to the node's extent
    auto blend = [&](f3 pos) {
        rgb = rgb + (1.f - a) * AlphaPerLev * pos;
        a = a + (1.f - a) * AlphaPerLev;
    };

    float tEnter, tExit;
    HitAABB(ray, AABB{F3(0.f), F3(1.f)}, tEnter, tExit);
    bool go = !(cull && tEnter >= tExit);
    if (go) {
        uint32_t maxStepCnt = 0;
        if (target == KK_VDB_TARGET_POSITIONS_AFTER_DEPTH_SKIP)
            maxStepCnt = static_cast<uint32_t>(
                ceilf(rp.sampling_rate * length((tExit - tEnter) * ray.dir) *
                      gmax(voxPerVol.x, gmax(voxPerVol.y, voxPerVol.z))));
        ray.pos = ray.pos * voxPerVol;
        ray.dir = normalize(ray.dir * voxPerVol);
        HitAABB(ray, AABB{F3(0.f), voxPerVol}, tEnter, tExit);
        go = !(cull && tEnter >= tExit);
        if (go)
            step = (tExit - tEnter) / (maxStepCnt - 1);
    }
    if (go) {
        HitAABB(ray, AABB{F3(0.f), F3(p.vox_per_vdb)}, tEnter, tExit);
        go = !(tEnter >= tExit);
    }
    if (go) {
        int lev = p.root_lev;
        float tExits[2] = {0.f, 0.f};
        kk_vdb_node nodes[2];
        nodes[0] = nodes[1] = vdb.node_pools[pick3(p.node_starts, p.root_lev)];
        tExits[lev - 1] = tExit - Eps;
        HDDA3D h;
        h.sign = {sgn(ray.dir.x), sgn(ray.dir.y), sgn(ray.dir.z)};
        h.mask = {0, 0, 0};
        h.tCurr = tEnter + Eps;
        h.tNext = 0.f;
        const f3 absInvDir = vabs(1.f / ray.dir);
        h.Prepare(ray, absInvDir, F3(0.f), pick3(p.voxs_per_child, p.root_lev));
        if (target == KK_VDB_TARGET_AABBS && lev == displayLev)
            blend((ray.pos + h.tCurr * ray.dir - toF3(nodes[lev - 1].pos)) / p.vox_per_vdb);

        for (int guard = 0; guard < (1 << 20) && lev != p.root_lev + 1; ++guard) {
            const int dimLev = p.dims[lev];
            if (h.chPos.x < 0 || h.chPos.x >= dimLev || h.chPos.y < 0 || h.chPos.y >= dimLev ||
                h.chPos.z < 0 || h.chPos.z >= dimLev)
                break;
            h.Next();
            const int l2 = p.log2_dims[lev];
            const uint32_t rel = (static_cast<uint32_t>(h.chPos.z) << (l2 << 1)) |
                                 (static_cast<uint32_t>(h.chPos.y) << l2) | static_cast<uint32_t>(h.chPos.x);
            const uint32_t ch =
                vdb.child_pools[p.child_starts[lev - 1] + nodes[lev - 1].child_pool_offs_relative + rel];
            if (ch == UndefRltIdx) {
                h.Step();
            } else if (lev > 1) {
                const float pushedExit = h.tNext - Eps;
                --lev;
                nodes[lev - 1] = vdb.node_pools[p.node_starts[lev] + ch];
                tExits[lev - 1] = pushedExit;
                h.tCurr += Eps;
                h.Prepare(ray, absInvDir, toF3(nodes[lev - 1].pos), p.voxs_per_child[lev]);
                if (target == KK_VDB_TARGET_AABBS && lev == displayLev)
                    blend((ray.pos + h.tCurr * ray.dir - toF3(nodes[lev - 1].pos)) /
                          p.voxs_per_child[lev + 1]);
            } else {
                h.tCurr += Eps;
                float tCurr = h.tCurr;
                const float tLeafExit = h.tNext - Eps;
                const kk_vdb_node leaf = vdb.node_pools[p.node_starts[0] + ch];
                const f3 brickMin = toF3(leaf.pos);
                const T *__restrict__ brick = atlas + brickVox * ch;
                bool done = false;
                if (target == KK_VDB_TARGET_AABBS) {
                    if (displayLev == 0)
                        blend((ray.pos + tCurr * ray.dir - brickMin) / dims0f);
                } else if (layer != displayLeafLayer) {
                    layer = (layer + 1) & 0xFF; // a uint8_t counter in the reference
                } else if (target == KK_VDB_TARGET_DEPTHS) {
                    const f3 pib = ray.pos + tCurr * ray.dir - brickMin;
                    DepthDDA2D d;
                    if (d.Init(tCurr, dims0f, p.min_dep_pos_val_in_brick, p.max_dep_pos_val_in_brick, pib, h.sign, absInvDir, ray)) {
                        const float depNorm = static_cast<float>(depthAt(brick, d.posInBrick)) / dims0f;
                        rgb = {depNorm, depNorm, depNorm};
                    } else
                        rgb = {1.f, 0.f, 0.f};
                    a = 1.f;
                    done = true;
                } else {
                    tCurr = step * ceilf(tCurr / step);
                    f3 pib = ray.pos + tCurr * ray.dir - brickMin;
                    bool skipped = false;
                    DepthDDA2D d;
                    if (d.Init(tCurr, dims0f, p.min_dep_pos_val_in_brick, p.max_dep_pos_val_in_brick, pib, h.sign, absInvDir, ray))
                        while (static_cast<float>(depthAt(brick, d.posInBrick)) > d.dep) {
                            if (d.t >= tLeafExit) {
                                skipped = true;
                                break;
                            }
                            tCurr = d.t;
                            d.StepNext();
                        }
                    if (!skipped) {
                        tCurr = step * ceilf(tCurr / step);
                        pib = ray.pos + tCurr * ray.dir - brickMin;
                        rgb = pib / dims0f;
                        a = 1.f;
                        done = true;
                    }
                }
                if (done)
                    break;
                h.Step();
            }
            while (h.tCurr >= tExits[lev - 1]) {
                ++lev;
                if (lev == p.root_lev + 1)
                    break;
                h.Prepare(ray, absInvDir, toF3(nodes[lev - 1].pos), p.voxs_per_child[lev]);
            }
        }
    }
    reinterpret_cast<uchar4 *>(rgba)[static_cast<size_t>(y) * W + x] =
        make_uchar4(satU8(255.f * rgb.x), satU8(255.f * rgb.y), satU8(255.f * rgb.z), satU8(255.f * a));
}

template <typename T>
int launchTarget(const kk_vdb_device_data *h_vdb, const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                 int target, int displayLev, int displayLeafLayer, uint8_t *d_rgba, int width, int height,
                 int y0, int y1, hipStream_t stream) {
    KK_REQUIRE(h_vdb && h_params && h_frame && d_rgba, "null pointer");
    KK_REQUIRE(h_vdb->node_pools && h_vdb->child_pools && h_vdb->atlas, "VDB arrays missing");
    KK_REQUIRE(h_vdb->voxel_bytes == sizeof(T), "voxel type does not match the VDB");
    KK_REQUIRE(h_vdb->vdb_param.root_lev >= 1 && h_vdb->vdb_param.root_lev <= 2, "root level out of range");
    KK_REQUIRE(target == KK_VDB_TARGET_AABBS || target == KK_VDB_TARGET_DEPTHS ||
                   target == KK_VDB_TARGET_POSITIONS_AFTER_DEPTH_SKIP,
               "unknown render target");
    KK_REQUIRE(displayLev >= 0 && displayLev <= 255 && displayLeafLayer >= 0 && displayLeafLayer <= 255,
               "display level / leaf layer out of range");
    KK_REQUIRE(width > 0 && height > 0 && y0 >= 0 && y1 <= height && y0 <= y1, "bad image size or rows");
    if (y0 == y1)
        return KK_OK;
    dim3 grid(div_up(width, 16), div_up(y1 - y0, 16));
    vdb_target_kernel<T><<<grid, THREADS, 0, stream>>>(*h_vdb, *h_params, *h_frame, target, displayLev,
                                                       displayLeafLayer, d_rgba, width, height, y0, y1);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

template <typename T>
int launchRaw(const T *d_volume, uint32_t dx, uint32_t dy, uint32_t dz, const float *d_tf,
              const kk_dvr_params *h_params, const kk_dvr_frame *h_frame, int target, uint8_t *d_rgba,
              int width, int height, int y0, int y1, unsigned long long *d_counters, hipStream_t stream) {
    KK_REQUIRE(d_volume && d_tf && h_params && h_frame && d_rgba, "null pointer");
    KK_REQUIRE(dx > 0 && dy > 0 && dz > 0 && dx < (1u << 30) && dy < (1u << 30) && dz < (1u << 30),
               "volume extent out of range");
    KK_REQUIRE(target == KK_RAW_TARGET_SCENE || target == KK_RAW_TARGET_AABB, "unknown render target");
    KK_REQUIRE(width > 0 && height > 0 && y0 >= 0 && y1 <= height && y0 <= y1, "bad image size or rows");
    if (y0 == y1)
        return KK_OK;
    dim3 grid(div_up(width, 16), div_up(y1 - y0, 16));
    if (d_counters)
        raw_kernel<T, true><<<grid, THREADS, 0, stream>>>(d_volume, static_cast<int>(dx), static_cast<int>(dy),
                                                          static_cast<int>(dz), d_tf, *h_params, *h_frame,
                                                          target, d_rgba, width, height, y0, y1, d_counters);
    else
        raw_kernel<T, false><<<grid, THREADS, 0, stream>>>(d_volume, static_cast<int>(dx), static_cast<int>(dy),
                                                           static_cast<int>(dz), d_tf, *h_params, *h_frame,
                                                           target, d_rgba, width, height, y0, y1, nullptr);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

template <typename T>
int launch(const kk_vdb_device_data *h_vdb, const float *d_tf, const kk_dvr_params *h_params,
           const kk_dvr_frame *h_frame, uint8_t *d_rgba, int width, int height, int y0, int y1,
           int stripeFirst, int stripeStride, unsigned long long *d_counters, hipStream_t stream,
           int compact = 0, uint32_t *d_tile_cost = nullptr, const uint32_t *d_tile_order = nullptr) {
    KK_REQUIRE(h_vdb && d_tf && h_params && h_frame && d_rgba, "null pointer");
    KK_REQUIRE(h_vdb->node_pools && h_vdb->child_pools && h_vdb->atlas, "VDB arrays missing");
    KK_REQUIRE(h_vdb->voxel_bytes == sizeof(T), "voxel type does not match the VDB");
    KK_REQUIRE(h_vdb->vdb_param.root_lev >= 1 && h_vdb->vdb_param.root_lev <= 2, "root level out of range");
    KK_REQUIRE(width > 0 && height > 0 && y0 >= 0 && y1 <= height && y0 <= y1, "bad image size or rows");
    KK_REQUIRE(stripeStride >= 1 && stripeFirst >= 0, "bad stripe selection");
    for (int lev = 1; lev <= h_vdb->vdb_param.root_lev; ++lev) {
        // the marcher scales by these instead of dividing, which is exact for powers of two only
        int e = 0;
        const float v = h_vdb->vdb_param.voxs_per_child[lev];
        KK_REQUIRE(v >= 1.f && v <= 1073741824.f && frexpf(v, &e) == .5f,
                   "voxels per child must be a power of two");
    }
    const int stripes = static_cast<int>(div_up(y1 - y0, 16));
    if (y0 == y1 || stripeFirst >= stripes)
        return KK_OK;
    dim3 grid(div_up(width, 16), div_up(stripes - stripeFirst, stripeStride));
    // the shading branch is a compile-time variant, like the reference's renderScene<T, ., UseShading>
#define KK_LAUNCH_DVR(COUNT, SHADING, AFIX)                                                        \
    dvr_kernel<T, COUNT, SHADING, AFIX><<<grid, THREADS, 0, stream>>>(                             \
        *h_vdb, d_tf, *h_params, *h_frame, d_rgba, width, height, y0, y1, stripeFirst, stripeStride, compact, \
        d_counters, d_tile_cost, d_tile_order)
    const bool shading = h_params->use_shading != 0;
    // The variant for the 36-voxel atlas brick makes a sample's offsets in floating point and in 32 bits: every
    // voxel's offset in a cube of edge 36 must stay below 2^24 and the atlas below 4 GiB.
    const kk_vdb_param &vp = h_vdb->vdb_param;
    const uint64_t cubeEnd = (static_cast<uint64_t>(vp.atlas_brick_per_vdb[2] > 0 ? vp.atlas_brick_per_vdb[2] : 0) * 36u * 36u +
                              static_cast<uint64_t>(vp.atlas_brick_per_vdb[1] > 0 ? vp.atlas_brick_per_vdb[1] : 0) * 36u) * 36u +
                             static_cast<uint64_t>(vp.atlas_brick_per_vdb[0] > 0 ? vp.atlas_brick_per_vdb[0] : 0) * 36u;
    const bool edge36 = vp.vox_per_atlas_brick == 36 && cubeEnd < (1ull << 24) &&
                        h_vdb->leaf_num * (36ull * 36ull * 36ull) * sizeof(T) < (1ull << 32);
    if (d_counters) {
        if (shading)
            KK_LAUNCH_DVR(true, true, 0);
        else
            KK_LAUNCH_DVR(true, false, 0);
    } else if (edge36) {
        if (shading)
            KK_LAUNCH_DVR(false, true, 36);
        else
            KK_LAUNCH_DVR(false, false, 36);
    } else {
        if (shading)
            KK_LAUNCH_DVR(false, true, 0);
        else
            KK_LAUNCH_DVR(false, false, 0);
    }
#undef KK_LAUNCH_DVR
    KK_LAUNCH_CHECK();
    return KK_OK;
}

// The cost of a tile in 256 classes, sixteen to a doubling: the place of its leading one and the four bits
// below it. A list schedule needs no finer order, and eight bits sort in one pass.
__device__ __forceinline__ uint32_t tile_cost_class(uint32_t cost) {
    // as a float the cost (< 2^24, so exact) has the place of its leading one in the exponent and the bits
    // below it at the top of the mantissa
    const uint32_t c = cost > 65535u ? 65535u : cost;
    return c ? (__float_as_uint(static_cast<float>(c)) >> 19) - (127u << 4) : 0u;
}
// Keys for the sort that puts the costly tiles first (the cost class, complemented), and the counts of
// their digit for the sort's pass (radix_hist.h): a few workgroups, each over a share of the tiles.
constexpr uint32_t TILE_KEY_GROUPS = 16;
__global__ __launch_bounds__(THREADS) void tile_keys_kernel(const uint32_t *__restrict__ cost, uint32_t n,
                                                            uint32_t *__restrict__ keys,
                                                            uint32_t *__restrict__ ghist) {
    __shared__ uint32_t s_hist[rsort::HIST_RADIX];
    rsort::hist_clear<1>(s_hist, threadIdx.x, THREADS);
    for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < n; i += gridDim.x * THREADS) {
        const uint32_t key = 255u - tile_cost_class(cost[i]);
        keys[i] = key;
        rsort::hist_add<1>(s_hist, key);
    }
    rsort::hist_flush<1>(s_hist, ghist, threadIdx.x, THREADS);
}

// After the sort: an order made by a sort whose look-back gave up is no permutation of the tiles (tiles
// would be left unrendered), so it is replaced by the launch order.
__global__ __launch_bounds__(THREADS) void tile_order_fixup_kernel(const uint32_t *__restrict__ sortError,
                                                                   uint32_t *__restrict__ order, uint32_t n) {
    if (*sortError == 0u)
        return;
    for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < n; i += gridDim.x * THREADS)
        order[i] = i;
}

// The whole order in one launch, for frames of up to ORDER_ONE_LAUNCH_MAX tiles: a counting sort over the 256
// classes in which no workgroup waits for another. A workgroup places 1,024 consecutive tiles and counts for
// itself what it has to know about the others: the classes of the tiles in front of its own and of those from
// its own on (the times of all tiles are a few hundred KB that stay in L2). A tile's place = the start of its
// class + the tiles of the class in front of the workgroup's + those in the waves in front of its own + those
// in lower lanes of its wave, so tiles of one class keep their order. Nothing can give up, so there is no
// error word to raise; the word the status call reads is cleared.
constexpr int ORDER_THREADS = 1024;
constexpr int ORDER_WAVES = ORDER_THREADS / 64;
constexpr uint32_t ORDER_ONE_LAUNCH_MAX = 1u << 18;
__global__ __launch_bounds__(ORDER_THREADS) void tile_order_kernel(const uint32_t *__restrict__ cost, uint32_t n,
                                                                   uint32_t *__restrict__ order,
                                                                   uint32_t *__restrict__ errorWord) {
    __shared__ uint32_t s_count[2][256]; // classes of the tiles in front of this workgroup's / from them on
    __shared__ uint32_t s_wave[ORDER_WAVES][256];
    __shared__ uint32_t s_tmp[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 256; i += ORDER_THREADS)
        (&s_count[0][0])[i] = 0u;
    for (int i = tid; i < ORDER_WAVES * 256; i += ORDER_THREADS)
        (&s_wave[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t rounds = (n + ORDER_THREADS - 1u) / ORDER_THREADS;
    uint32_t myKey = 0u;
    for (uint32_t r0 = 0; r0 < rounds; r0 += 4u) { // four loads in flight per thread
        uint32_t c[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t i = (r0 + u) * ORDER_THREADS + tid;
            c[u] = i < n ? cost[i] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t r = r0 + u, i = r * ORDER_THREADS + tid;
            const bool in = i < n;
            const uint32_t key = 255u - tile_cost_class(c[u]);
            if (r == blockIdx.x)
                myKey = key;
            // Most tiles of a frame are empty space and share a class: the lanes whose class is lane 0's are
            // counted with one addition instead of one each on the same word.
            const uint32_t first = __builtin_amdgcn_readfirstlane(key);
            const unsigned long long same = __ballot(in && key == first);
            uint32_t *row = s_count[r < blockIdx.x ? 0 : 1];
            if (lane == 0 && same != 0ull)
                atomicAdd(&row[first], static_cast<uint32_t>(__popcll(same)));
            if (in && key != first)
                atomicAdd(&row[key], 1u);
        }
    }
    // this workgroup's tiles: rank among the tiles of the same class in lower lanes of the wave
    const uint32_t mine = blockIdx.x * ORDER_THREADS + tid;
    const bool valid = mine < n;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (myKey >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    const uint32_t below = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && below == 0u) // the lowest lane of a class counts it for the wave
        s_wave[wave][myKey] = __popcll(peers);
    __syncthreads();
    if (tid < 256) {
        const uint32_t front = s_count[0][tid], total = front + s_count[1][tid];
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if (lane >= o)
                incl += t;
        }
        if (lane == 63)
            s_tmp[wave] = incl;
        s_count[1][tid] = incl - total + front; // the class's start (less the scanning waves in front) + its tiles in front
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t run = s_count[1][tid];
        for (int w = 0; w < wave; ++w)
            run += s_tmp[w];
        for (int w = 0; w < ORDER_WAVES; ++w) {
            const uint32_t cnt = s_wave[w][tid];
            s_wave[w][tid] = run;
            run += cnt;
        }
    }
    __syncthreads();
    if (valid)
        order[s_wave[wave][myKey] + below] = mine;
    if (blockIdx.x == 0 && tid == 0)
        *errorWord = 0u;
}

struct OrderWs {
    size_t keysA, keysB, sortWs, sortWsBytes, total;
};
inline OrderWs order_ws(uint32_t tiles) {
    OrderWs w;
    Carver c;
    w.keysA = c.take(4 * static_cast<size_t>(tiles));
    w.keysB = c.take(4 * static_cast<size_t>(tiles));
    w.sortWsBytes = kk_sort_workspace_bytes(tiles);
    w.sortWs = c.take(w.sortWsBytes);
    w.total = c.off;
    return w;
}

} // namespace dvr
} // namespace kk

extern "C" uint32_t kk_dvr_tile_count(int width, int rows) {
    return width > 0 && rows > 0 ? kk::div_up(width, 16) * kk::div_up(rows, 16) : 0u;
}
extern "C" size_t kk_dvr_tile_order_workspace_bytes(uint32_t tiles) { return kk::dvr::order_ws(tiles).total; }
extern "C" int kk_dvr_tile_order(const uint32_t *d_tile_cost, uint32_t tiles, uint32_t *d_tile_order,
                                 void *d_workspace, size_t workspace_bytes, void *stream_) {
    using namespace kk;
    using namespace kk::dvr;
    KK_REQUIRE(d_tile_cost && d_tile_order && d_workspace, "null pointer");
    KK_REQUIRE(tiles > 0 && tiles <= KK_LBVH_MAX_FACES, "tile count out of range");
    const OrderWs w = order_ws(tiles);
    if (workspace_bytes < w.total) {
        set_error("tile order workspace too small: %zu < %zu", workspace_bytes, w.total);
        return KK_ERR_WORKSPACE;
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *b = static_cast<char *>(d_workspace);
    if (tiles <= ORDER_ONE_LAUNCH_MAX && !rsort::force_error()) {
        uint32_t *errorWord = reinterpret_cast<uint32_t *>(b + w.sortWs + kk_sort_workspace_error_offset(tiles));
        tile_order_kernel<<<div_up(tiles, ORDER_THREADS), ORDER_THREADS, 0, stream>>>(d_tile_cost, tiles,
                                                                                      d_tile_order, errorWord);
        KK_LAUNCH_CHECK();
        return KK_OK;
    }
    // larger frames: keys, one pass of the radix sort, and the launch order should its look-back give up
    uint32_t *keysA = reinterpret_cast<uint32_t *>(b + w.keysA);
    uint32_t *ghist = nullptr;
    const int rc = rsort::sort_prepare(b + w.sortWs, w.sortWsBytes, tiles, stream, &ghist);
    if (rc != KK_OK)
        return rc;
    const uint32_t grid = div_up(tiles, THREADS) < TILE_KEY_GROUPS ? div_up(tiles, THREADS) : TILE_KEY_GROUPS;
    tile_keys_kernel<<<grid, THREADS, 0, stream>>>(d_tile_cost, tiles, keysA, ghist);
    KK_LAUNCH_CHECK();
    // one 8-bit pass from the A arrays (whose values are the tile numbers themselves) into the B arrays
    uint32_t *sortError = nullptr;
    const int rc2 = rsort::sort_pairs(keysA, reinterpret_cast<uint32_t *>(b + w.keysB), nullptr, d_tile_order, true,
                                      tiles, b + w.sortWs, w.sortWsBytes, stream, &sortError, 1, true);
    if (rc2 != KK_OK)
        return rc2;
    tile_order_fixup_kernel<<<grid, THREADS, 0, stream>>>(sortError, d_tile_order, tiles);
    KK_LAUNCH_CHECK();
    return KK_OK;
}
extern "C" int kk_dvr_tile_order_status(const void *d_workspace, uint32_t tiles, void *stream_) {
    using namespace kk;
    KK_REQUIRE(d_workspace != nullptr, "null pointer");
    KK_REQUIRE(tiles > 0 && tiles <= KK_LBVH_MAX_FACES, "tile count out of range");
    uint32_t word = 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    KK_HIP_TRY(hipMemcpyAsync(&word,
                              static_cast<const char *>(d_workspace) + dvr::order_ws(tiles).sortWs +
                                  kk_sort_workspace_error_offset(tiles),
                              sizeof(word), hipMemcpyDeviceToHost, stream));
    KK_HIP_TRY(hipStreamSynchronize(stream));
    if (word != 0) {
        set_error("kk_dvr_tile_order: the sort's look-back gave up (error word %u); the launch order was kept",
                  word);
        return KK_ERR_DEVICE;
    }
    return KK_OK;
}
#define KK_DVR_ORDERED(SUFFIX, TYPE)                                                                          \
    extern "C" int kk_dvr_render_ordered_##SUFFIX(                                                            \
        const kk_vdb_device_data *h_vdb, const float *d_tf, const kk_dvr_params *h_params,                    \
        const kk_dvr_frame *h_frame, uint8_t *d_rgba, int width, int height, int y0, int y1,                  \
        uint32_t *d_tile_cost, const uint32_t *d_tile_order, void *stream) {                                  \
        return kk::dvr::launch<TYPE>(h_vdb, d_tf, h_params, h_frame, d_rgba, width, height, y0, y1, 0, 1, nullptr, \
                                     static_cast<hipStream_t>(stream), 0, d_tile_cost, d_tile_order);         \
    }
KK_DVR_ORDERED(u8, uint8_t)
KK_DVR_ORDERED(u16, uint16_t)
#undef KK_DVR_ORDERED

extern "C" int kk_dvr_render_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                uint8_t *d_rgba, int width, int height, int y0, int y1,
                                unsigned long long *d_counters, void *stream) {
    return kk::dvr::launch<uint8_t>(h_vdb, d_tf, h_params, h_frame, d_rgba, width, height, y0, y1, 0, 1,
                                    d_counters, static_cast<hipStream_t>(stream));
}
extern "C" int kk_dvr_render_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                 const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                 uint8_t *d_rgba, int width, int height, int y0, int y1,
                                 unsigned long long *d_counters, void *stream) {
    return kk::dvr::launch<uint16_t>(h_vdb, d_tf, h_params, h_frame, d_rgba, width, height, y0, y1, 0,
                                     1, d_counters, static_cast<hipStream_t>(stream));
}
extern "C" int kk_dvr_render_stripes_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                        const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                        uint8_t *d_rgba, int width, int height, int stripe_first,
                                        int stripe_stride, unsigned long long *d_counters,
                                        void *stream) {
    return kk::dvr::launch<uint8_t>(h_vdb, d_tf, h_params, h_frame, d_rgba, width, height, 0, height,
                                    stripe_first, stripe_stride, d_counters,
                                    static_cast<hipStream_t>(stream));
}
extern "C" int kk_dvr_render_stripes_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                         const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                         uint8_t *d_rgba, int width, int height, int stripe_first,
                                         int stripe_stride, unsigned long long *d_counters,
                                         void *stream) {
    return kk::dvr::launch<uint16_t>(h_vdb, d_tf, h_params, h_frame, d_rgba, width, height, 0, height,
                                     stripe_first, stripe_stride, d_counters,
                                     static_cast<hipStream_t>(stream));
}
extern "C" int kk_dvr_render_stripes_compact_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                                const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                                uint8_t *d_stripes, int width, int height, int stripe_first,
                                                int stripe_stride, unsigned long long *d_counters,
                                                void *stream) {
    return kk::dvr::launch<uint8_t>(h_vdb, d_tf, h_params, h_frame, d_stripes, width, height, 0, height,
                                    stripe_first, stripe_stride, d_counters,
                                    static_cast<hipStream_t>(stream), 1);
}
extern "C" int kk_dvr_render_stripes_compact_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                                 const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                                 uint8_t *d_stripes, int width, int height, int stripe_first,
                                                 int stripe_stride, unsigned long long *d_counters,
                                                 void *stream) {
    return kk::dvr::launch<uint16_t>(h_vdb, d_tf, h_params, h_frame, d_stripes, width, height, 0, height,
                                     stripe_first, stripe_stride, d_counters,
                                     static_cast<hipStream_t>(stream), 1);
}
// the compact stripes of one GPU with the tile order and tile times of kk_dvr_render_ordered_*
extern "C" uint32_t kk_dvr_stripe_tile_count(int width, int height, int stripe_first, int stripe_stride) {
    if (width <= 0 || height <= 0 || stripe_first < 0 || stripe_stride < 1)
        return 0u;
    const int stripes = static_cast<int>(kk::div_up(height, 16));
    return stripe_first >= stripes ? 0u : kk::div_up(width, 16) * kk::div_up(stripes - stripe_first, stripe_stride);
}
#define KK_DVR_STRIPES_ORDERED(SUFFIX, TYPE)                                                                  \
    extern "C" int kk_dvr_render_stripes_compact_ordered_##SUFFIX(                                            \
        const kk_vdb_device_data *h_vdb, const float *d_tf, const kk_dvr_params *h_params,                    \
        const kk_dvr_frame *h_frame, uint8_t *d_stripes, int width, int height, int stripe_first,             \
        int stripe_stride, uint32_t *d_tile_cost, const uint32_t *d_tile_order, void *stream) {               \
        return kk::dvr::launch<TYPE>(h_vdb, d_tf, h_params, h_frame, d_stripes, width, height, 0, height,     \
                                     stripe_first, stripe_stride, nullptr, static_cast<hipStream_t>(stream),  \
                                     1, d_tile_cost, d_tile_order);                                           \
    }
KK_DVR_STRIPES_ORDERED(u8, uint8_t)
KK_DVR_STRIPES_ORDERED(u16, uint16_t)
#undef KK_DVR_STRIPES_ORDERED
extern "C" int kk_raw_render_u8(const uint8_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                                const float *d_tf, const kk_dvr_params *h_params,
                                const kk_dvr_frame *h_frame, int target, uint8_t *d_rgba, int width,
                                int height, int y0, int y1, unsigned long long *d_counters, void *stream) {
    return kk::dvr::launchRaw<uint8_t>(d_volume, dx, dy, dz, d_tf, h_params, h_frame, target, d_rgba, width,
                                       height, y0, y1, d_counters, static_cast<hipStream_t>(stream));
}
extern "C" int kk_raw_render_u16(const uint16_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                                 const float *d_tf, const kk_dvr_params *h_params,
                                 const kk_dvr_frame *h_frame, int target, uint8_t *d_rgba, int width,
                                 int height, int y0, int y1, unsigned long long *d_counters, void *stream) {
    return kk::dvr::launchRaw<uint16_t>(d_volume, dx, dy, dz, d_tf, h_params, h_frame, target, d_rgba, width,
                                        height, y0, y1, d_counters, static_cast<hipStream_t>(stream));
}
extern "C" int kk_dvr_render_target_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                       const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                       int target, int display_lev, int display_leaf_layer,
                                       uint8_t *d_rgba, int width, int height, int y0, int y1, void *stream) {
    if (target == KK_VDB_TARGET_SCENE)
        return kk::dvr::launch<uint8_t>(h_vdb, d_tf, h_params, h_frame, d_rgba, width, height, y0, y1, 0, 1,
                                        nullptr, static_cast<hipStream_t>(stream));
    return kk::dvr::launchTarget<uint8_t>(h_vdb, h_params, h_frame, target, display_lev, display_leaf_layer,
                                          d_rgba, width, height, y0, y1, static_cast<hipStream_t>(stream));
}
extern "C" int kk_dvr_render_target_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                        const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                        int target, int display_lev, int display_leaf_layer,
                                        uint8_t *d_rgba, int width, int height, int y0, int y1, void *stream) {
    if (target == KK_VDB_TARGET_SCENE)
        return kk::dvr::launch<uint16_t>(h_vdb, d_tf, h_params, h_frame, d_rgba, width, height, y0, y1, 0,
                                         1, nullptr, static_cast<hipStream_t>(stream));
    return kk::dvr::launchTarget<uint16_t>(h_vdb, h_params, h_frame, target, display_lev, display_leaf_layer,
                                           d_rgba, width, height, y0, y1, static_cast<hipStream_t>(stream));
}
