// Depth-Box VDB build: DepthBoxVDB::BuildFrom<T> (src/ray_caster/dpbx_vdb.cu:275-610).
//   classify   (device) one workgroup per brick of the leaf grid reads its dims0^3 voxels with
//              row-contiguous loads and decides whether the brick is empty under the transfer function
//              (dpbx_vdb.cu:345-385, where one thread walks the brick through the texture unit);
//   plan, wire (host)   tree parameters, node pools sorted by (level, x, y, z), dense child pools, atlas
//              map (dpbx_vdb.cu:282-338, 387-571). A few thousand nodes: not worth a kernel;
//   fill_atlas (device) one workgroup per valid brick writes its (dims0 + 2 apronDepWid)^3 atlas voxels
//              contiguously: volume data inside, depth values on the one-face shell (dpbx_vdb.cu:176-272).
// Point sampling of the volume (tex3D, cudaFilterModePoint, border addressing) is "voxel at the integer
// coordinate, 0 outside the volume".
#include "kk_host.h"

#include <algorithm>
#include <string.h>
#include <vector>

namespace kk {
namespace vdb {

constexpr int THREADS = 256;
constexpr int MAX_RNGS = 8;

// The empty scalar ranges as kernel arguments: lo | hi << 16 per range. Kernels read them at constant
// positions (loops over the ranges are unrolled and leave at the count, which is the same for every
// lane), so they stay in scalar registers; indexed with a run-time position they were fetched from
// memory again for every voxel.
template <typename T> struct Ranges {
    uint32_t b[MAX_RNGS];
    uint32_t n;
};

template <typename T>
__device__ __forceinline__ T voxelAt(const T *__restrict__ vol, int dx, int dy, int dz, int x, int y,
                                     int z) {
    if (x < 0 || y < 0 || z < 0 || x >= dx || y >= dy || z >= dz)
        return 0;
    return vol[(static_cast<size_t>(z) * dy + y) * dx + x];
}

// number of ranges that hold v (the reference counts (voxel, range) pairs, dpbx_vdb.cu:345-385)
template <typename T> __device__ __forceinline__ uint32_t rangeHits(T v, const Ranges<T> &r) {
    uint32_t c = 0;
    const uint32_t x = v;
#pragma unroll
    for (uint32_t i = 0; i < MAX_RNGS; ++i) {
        if (i >= r.n)
            break;
        c += (x >= (r.b[i] & 0xffffu) && x <= (r.b[i] >> 16)) ? 1u : 0u;
    }
    return c;
}

// Wide form of the classification for rows that can be read in aligned 16-byte pieces (dx a multiple
// of 16 / sizeof(T), 16-byte aligned volume): one workgroup takes GROUP bricks that are neighbours
// along x, so that every 128-byte line of the volume is fetched once and used whole (a single brick's
// 32-byte rows would each pull a whole line, four times over the kernel).
template <typename T, int GROUP>
__global__ __launch_bounds__(THREADS) void classify_wide_kernel(const T *__restrict__ vol, int dx, int dy,
                                                                int dz, int log2B, uint32_t lx,
                                                                uint32_t ly, Ranges<T> rngs,
                                                                uint32_t *__restrict__ leafValid) {
    constexpr int PER = 16 / sizeof(T); // voxels per 16-byte piece
    __shared__ uint32_t s_hits[GROUP];
    const int B = 1 << log2B;
    const uint32_t groupsX = (lx + GROUP - 1) / GROUP;
    const uint32_t gx = blockIdx.x % groupsX, gy = (blockIdx.x / groupsX) % ly, gz = blockIdx.x / (groupsX * ly);
    const int x0 = static_cast<int>(gx * GROUP) << log2B, y0 = static_cast<int>(gy) << log2B,
              z0 = static_cast<int>(gz) << log2B;
    if (threadIdx.x < GROUP)
        s_hits[threadIdx.x] = 0;
    __syncthreads();
    const int piecesPerRow = GROUP * B / PER, piecesPerBrick = B / PER;
    const int pieceNum = piecesPerRow * B * B;
    // a thread's pieces all belong to one brick when THREADS is a multiple of piecesPerRow
    uint32_t hits[GROUP];
#pragma unroll
    for (int g = 0; g < GROUP; ++g)
        hits[g] = 0;
    for (int p = threadIdx.x; p < pieceNum; p += THREADS) {
        const int px = p % piecesPerRow, row = p / piecesPerRow;
        const int y = y0 + (row & (B - 1)), z = z0 + (row >> log2B);
        const int x = x0 + px * PER;
        const int g = px / piecesPerBrick;
        uint32_t h = 0;
        if (y < dy && z < dz && x < dx) { // dx is a multiple of PER: a piece is inside or outside
            const uint4 w = *reinterpret_cast<const uint4 *>(vol + (static_cast<size_t>(z) * dy + y) * dx + x);
            const T *v = reinterpret_cast<const T *>(&w);
#pragma unroll
            for (uint32_t k = 0; k < MAX_RNGS; ++k) { // one range against the whole piece at a time
                if (k >= rngs.n)
                    break;
                const uint32_t lo = rngs.b[k] & 0xffffu, hi = rngs.b[k] >> 16;
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    h += (static_cast<uint32_t>(v[i]) >= lo && static_cast<uint32_t>(v[i]) <= hi) ? 1u : 0u;
            }
        } else
            h = PER * rangeHits(static_cast<T>(0), rngs); // outside the volume reads as 0
#pragma unroll
        for (int gg = 0; gg < GROUP; ++gg)
            hits[gg] += gg == g ? h : 0u;
    }
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
        uint32_t h = hits[g];
        for (int o = 32; o > 0; o >>= 1)
            h += __shfl_xor(h, o, 64);
        if ((threadIdx.x & 63) == 0 && h != 0)
            atomicAdd(&s_hits[g], h);
    }
    __syncthreads();
    if (threadIdx.x < GROUP) {
        const uint32_t bx = gx * GROUP + threadIdx.x;
        if (bx < lx) {
            const uint32_t voxNum = 1u << (3 * log2B);
            leafValid[(static_cast<size_t>(gz) * ly + gy) * lx + bx] = s_hits[threadIdx.x] == voxNum ? 0u : 1u;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(THREADS) void classify_kernel(const T *__restrict__ vol, int dx, int dy,
                                                           int dz, int log2B, uint32_t lx, uint32_t ly,
                                                           Ranges<T> rngs,
                                                           uint32_t *__restrict__ leafValid) {
    __shared__ uint32_t s_red[THREADS / 64];
    const uint32_t leaf = blockIdx.x;
    const uint32_t lz_ = leaf / (lx * ly);
    const uint32_t ly_ = (leaf - lz_ * lx * ly) / lx;
    const uint32_t lx_ = leaf - lz_ * lx * ly - ly_ * lx;
    const int B = 1 << log2B;
    const int px = lx_ << log2B, py = ly_ << log2B, pz = lz_ << log2B;
    const uint64_t voxNum = 1ull << (3 * log2B);
    uint32_t hits = 0;
    for (uint64_t v = threadIdx.x; v < voxNum; v += THREADS) {
        const int x = static_cast<int>(v & (B - 1));
        const int y = static_cast<int>((v >> log2B) & (B - 1));
        const int z = static_cast<int>(v >> (2 * log2B));
        hits += rangeHits(voxelAt(vol, dx, dy, dz, px + x, py + y, pz + z), rngs);
    }
    for (int o = 32; o > 0; o >>= 1)
        hits += __shfl_xor(hits, o, 64);
    if ((threadIdx.x & 63) == 0)
        s_red[threadIdx.x >> 6] = hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        leafValid[leaf] = (total == static_cast<uint32_t>(voxNum)) ? 0u : 1u;
    }
}

// The same fill for bricks whose padded edge fits a 64-bit mask (A <= 64, every configuration the
// reference's applications use): the brick's A^3 voxels are staged in LDS once; one bit per voxel
// ("not in an empty scalar range") is packed into rows along x, y and z, so the inward march of a
// depth-shell voxel - "first position whose 3 x 3 window holds a non-empty voxel" - is the OR of nine
// row masks and a count of trailing / leading zeros.
constexpr int FILL_THREADS = 1024; // one workgroup per brick: many threads hide the latency of its loads
// AFIX: the padded brick edge as a compile-time constant (36 = 32-voxel leaves with a two-voxel apron, the
// layout of the reference's data sets), so that the divisions that turn a voxel index into coordinates
// are multiplications; 0: any edge.
template <typename T, int AFIX>
__global__ __launch_bounds__(FILL_THREADS) void fill_atlas_lds_kernel(const T *__restrict__ vol, int dx,
                                                                 int dy, int dz,
                                                                 const kk_vdb_node *__restrict__ leaves,
                                                                 int B, int apronWid, int apronDepWid,
                                                                 Ranges<T> rngs, T *__restrict__ atlas) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    const int A = AFIX ? AFIX : B + 2 * apronDepWid;
    const int A2 = A * A;
    const size_t brickVox = static_cast<size_t>(A) * A2;
    unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(s_raw); // [3][A][A]
    T *s_vox = reinterpret_cast<T *>(s_raw + sizeof(unsigned long long) * 3 * A2);
    const kk_vdb_node leaf = leaves[blockIdx.x];
    T *__restrict__ out = atlas + brickVox * blockIdx.x;
    const bool depthBox = apronDepWid != apronWid;
    const int tid = threadIdx.x;
    // Voxels move in 4-byte pieces where a row holds a whole number of them (A = 36, 20, 12 ... with
    // depth boxes): PER voxels per load, store and thread pass. The source pieces start two bytes off a
    // 4-byte boundary, which the memory system takes.
    constexpr int PER = 4 / sizeof(T);
    const bool pieces = A % PER == 0;
    if (pieces) {
        const int rowPieces = A / PER;
        uint32_t *s_vox4 = reinterpret_cast<uint32_t *>(s_vox);
        for (int g = tid; g < static_cast<int>(brickVox) / PER; g += FILL_THREADS) {
            const int j = g % rowPieces, ay = (g / rowPieces) % A, az = g / (rowPieces * A);
            const int x = leaf.pos[0] + j * PER - apronDepWid, y = leaf.pos[1] + ay - apronDepWid,
                      z = leaf.pos[2] + az - apronDepWid;
            uint32_t w = 0;
            if (y >= 0 && y < dy && z >= 0 && z < dz) {
                const T *src = vol + (static_cast<size_t>(z) * dy + y) * dx;
                if (x >= 0 && x + PER <= dx)
                    __builtin_memcpy(&w, src + x, 4);
                else
                    for (int i = 0; i < PER; ++i)
                        if (x + i >= 0 && x + i < dx)
                            w |= static_cast<uint32_t>(src[x + i]) << (8 * sizeof(T) * i);
            }
            s_vox4[g] = w;
        }
    } else
        for (int v = tid; v < static_cast<int>(brickVox); v += FILL_THREADS) {
            const int ax = v % A, ay = (v / A) % A, az = v / A2;
            s_vox[v] = voxelAt(vol, dx, dy, dz, leaf.pos[0] + ax - apronDepWid, leaf.pos[1] + ay - apronDepWid,
                               leaf.pos[2] + az - apronDepWid);
        }
    __syncthreads();
    if (!depthBox) {
        for (int v = tid; v < static_cast<int>(brickVox); v += FILL_THREADS)
            out[v] = s_vox[v];
        return;
    }
    // row r of direction d: bit i = voxel i along d is not empty. Rows are indexed [outer][inner]:
    // along x (outer z, inner y), along y (outer z, inner x), along z (outer y, inner x).
    // Every voxel is tested once, by the thread that owns its x row (z, y) and keeps that row's mask in a
    // register. Its bit also belongs to one y row and one z row: the z row (y, x) gets it by an LDS
    // atomic (the lanes of a wave hold different y, so different words), the y row (z, x) from the
    // wave's ballot - the lanes that share a z are consecutive y, so their ballot bits, shifted, are a
    // piece of that row, written by the first of them.
    for (int i = tid; i < 2 * A2; i += FILL_THREADS)
        s_mask[A2 + i] = 0ull;
    __syncthreads();
    const int lane = tid & 63;
    for (int r0 = 0; r0 < A2; r0 += FILL_THREADS) { // (whole waves: the ballot needs every lane)
        const int r = r0 + tid;
        const bool on = r < A2;
        const int az = on ? r / A : 0, ay = on ? r % A : 0;
        const bool leads = on && (ay == 0 || lane == 0); // first lane of the run of lanes with this z
        const int runLen = (A - ay) < (64 - lane) ? (A - ay) : (64 - lane);
        const unsigned long long runBits = runLen >= 64 ? ~0ull : ((1ull << runLen) - 1ull);
        unsigned long long m = 0;
        for (int ax = 0; ax < A; ++ax) {
            const bool bit = on && rangeHits(s_vox[r * A + ax], rngs) == 0;
            m |= static_cast<unsigned long long>(bit) << ax;
            if (bit)
                atomicOr(&s_mask[2 * A2 + ay * A + ax], 1ull << az);
            const unsigned long long seg = (__ballot(bit) >> lane) & runBits;
            if (leads && seg != 0ull)
                atomicOr(&s_mask[A2 + az * A + ax], seg << ay);
        }
        if (on)
            s_mask[r] = m;
    }
    __syncthreads();
    const unsigned long long field = (1ull << (B - 1)) - 1ull; // the B - 1 positions a march can stop at
    const int group = pieces ? PER : 1; // voxels a thread makes per pass, stored as one piece
    for (int v0 = tid * group; v0 < static_cast<int>(brickVox); v0 += FILL_THREADS * group) {
        uint32_t packed = 0;
        for (int sub = 0; sub < group; ++sub) {
            const int v = v0 + sub;
            const int ax = v % A, ay = (v / A) % A, az = v / A2;
            const int dirX = ax == 0 ? 1 : ax == A - 1 ? -1 : 0;
            const int dirY = ay == 0 ? 1 : ay == A - 1 ? -1 : 0;
            const int dirZ = az == 0 ? 1 : az == A - 1 ? -1 : 0;
            const int faces = (dirX != 0) + (dirY != 0) + (dirZ != 0);
            T value;
            if (faces == 0)
                value = s_vox[v];
            else if (faces > 1)
                value = 0;
            else {
                // the 3 x 3 window spans the two other axes; (o, i) = the centre's row coordinates
                const int d = dirX != 0 ? 0 : dirY != 0 ? 1 : 2;
                const int o = d == 2 ? ay : az, i = d == 0 ? ay : ax;
                const int dir = dirX + dirY + dirZ;
                const unsigned long long *rows = s_mask + d * A2;
                unsigned long long w = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    w |= rows[(o + t / 3 - 1) * A + (i + t % 3 - 1)];
                int dep = B - 1;
                if (dir > 0) { // from brick position 0 upwards: positions 0 .. B - 2
                    const unsigned long long f = (w >> apronDepWid) & field;
                    if (f)
                        dep = __ffsll(static_cast<long long>(f)) - 1;
                } else { // from brick position B - 1 downwards: positions B - 1 .. 1
                    const unsigned long long f = (w >> (apronDepWid + 1)) & field;
                    if (f)
                        dep = (B - 2) - (63 - __clzll(static_cast<long long>(f)));
                }
                value = static_cast<T>(dep);
            }
            if (pieces)
                packed |= static_cast<uint32_t>(value) << (8 * sizeof(T) * sub);
            else
                out[v] = value;
        }
        if (pieces) // a brick starts on a 4-byte boundary when its voxel count is a multiple of PER
            reinterpret_cast<uint32_t *>(out)[v0 / PER] = packed;
    }
}

template <typename T>
__global__ __launch_bounds__(THREADS) void fill_atlas_kernel(const T *__restrict__ vol, int dx, int dy,
                                                             int dz, const kk_vdb_node *__restrict__ leaves,
                                                             int B, int apronWid, int apronDepWid,
                                                             Ranges<T> rngs, T *__restrict__ atlas) {
    const int A = B + 2 * apronDepWid;
    const size_t brickVox = static_cast<size_t>(A) * A * A;
    const kk_vdb_node leaf = leaves[blockIdx.x];
    T *__restrict__ out = atlas + brickVox * blockIdx.x;
    const bool depthBox = apronDepWid != apronWid;
    for (size_t v = threadIdx.x; v < brickVox; v += THREADS) {
        const int ax = static_cast<int>(v % A);
        const int ay = static_cast<int>((v / A) % A);
        const int az = static_cast<int>(v / (static_cast<size_t>(A) * A));
        int dirX = 0, dirY = 0, dirZ = 0;
        if (depthBox) {
            dirX = ax == 0 ? 1 : ax == A - 1 ? -1 : 0;
            dirY = ay == 0 ? 1 : ay == A - 1 ? -1 : 0;
            dirZ = az == 0 ? 1 : az == A - 1 ? -1 : 0;
        }
        const int faces = (dirX != 0) + (dirY != 0) + (dirZ != 0);
        T value;
        if (faces == 0) {
            value = voxelAt(vol, dx, dy, dz, leaf.pos[0] + ax - apronDepWid, leaf.pos[1] + ay - apronDepWid,
                            leaf.pos[2] + az - apronDepWid);
        } else if (faces > 1) {
            value = 0;
        } else {
            // march inwards from the face with a 3 x 3 window until a non-empty voxel shows up
            int cx = leaf.pos[0] + (dirX == 1 ? 0 : dirX == -1 ? B - 1 : ax - apronDepWid);
            int cy = leaf.pos[1] + (dirY == 1 ? 0 : dirY == -1 ? B - 1 : ay - apronDepWid);
            int cz = leaf.pos[2] + (dirZ == 1 ? 0 : dirZ == -1 ? B - 1 : az - apronDepWid);
            // the two axes spanning the window
            const int ux = dirX != 0 ? 0 : dirY != 0 ? 0 : 1, uy = dirX != 0 ? 1 : 0, uz = dirY != 0 ? 1 : 0;
            const int wx = dirX != 0 ? 0 : dirY != 0 ? 1 : 0, wy = dirZ != 0 ? 1 : 0, wz = dirX != 0 ? 1 : 0;
            int dep;
            for (dep = 0; dep < B - 1; ++dep) {
                bool allEmpty = true;
                for (int t = 0; t < 9 && allEmpty; ++t) {
                    const int d0 = (t % 3) - 1, d1 = (t / 3) - 1;
                    const T s = voxelAt(vol, dx, dy, dz, cx + d0 * ux + d1 * wx, cy + d0 * uy + d1 * wy,
                                        cz + d0 * uz + d1 * wz);
                    allEmpty = rangeHits(s, rngs) != 0;
                }
                if (!allEmpty)
                    break;
                cx += dirX;
                cy += dirY;
                cz += dirZ;
            }
            value = static_cast<T>(dep);
        }
        out[v] = value;
    }
}

template <typename T> int makeRanges(const T *h, uint32_t n, Ranges<T> &r) {
    KK_REQUIRE(n <= MAX_RNGS, "too many empty scalar ranges");
    KK_REQUIRE(n == 0 || h != nullptr, "null pointer");
    r.n = n;
    for (uint32_t i = 0; i < MAX_RNGS; ++i)
        r.b[i] = i < n ? static_cast<uint32_t>(h[2 * i]) | (static_cast<uint32_t>(h[2 * i + 1]) << 16) : 0xffffu;
    return KK_OK;
}

template <typename T>
int classify(const T *d_volume, uint32_t dx, uint32_t dy, uint32_t dz, uint32_t log2B, const T *h_rngs,
             uint32_t n, uint32_t *d_leaf_valid, void *stream) {
    KK_REQUIRE(d_volume && d_leaf_valid, "null pointer");
    KK_REQUIRE(dx > 0 && dy > 0 && dz > 0, "empty volume");
    KK_REQUIRE(log2B >= 1 && log2B <= 10, "log2 dimension out of range");
    Ranges<T> r;
    int rc = makeRanges(h_rngs, n, r);
    if (rc != KK_OK)
        return rc;
    const uint32_t B = 1u << log2B;
    const uint32_t lx = (dx + B - 1) / B, ly = (dy + B - 1) / B, lz = (dz + B - 1) / B;
    constexpr uint32_t GROUP = 4, PER = 16 / sizeof(T);
    const bool wide = dx % PER == 0 && (reinterpret_cast<uintptr_t>(d_volume) & 15u) == 0 && B >= PER &&
                      log2B <= 6;
    if (wide)
        classify_wide_kernel<T, GROUP><<<((lx + GROUP - 1) / GROUP) * ly * lz, THREADS, 0,
                                         static_cast<hipStream_t>(stream)>>>(d_volume, dx, dy, dz, log2B,
                                                                            lx, ly, r, d_leaf_valid);
    else
        classify_kernel<T><<<lx * ly * lz, THREADS, 0, static_cast<hipStream_t>(stream)>>>(
            d_volume, dx, dy, dz, log2B, lx, ly, r, d_leaf_valid);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

// The leaf nodes' positions made on the GPU from the validity flags, so that the atlas can be filled while
// the host still wires the pools: leaf k is the k-th valid brick in (x, y, z) order, x slowest (the order
// of the pools, dpbx_vdb.cu:40-51). One workgroup: every thread counts the valid bricks of its share of
// that order, a scan over the threads gives its first k, a second walk writes the positions. The other
// fields of the nodes are left alone (the atlas fill reads `pos` only).
constexpr int LEAF_NODES_THREADS = 1024;
constexpr uint32_t LEAF_NODES_STAGED = 48 * 1024;
__global__ __launch_bounds__(LEAF_NODES_THREADS) void leaf_nodes_kernel(const uint32_t *__restrict__ leafValid,
                                                                      uint32_t lx, uint32_t ly, uint32_t lz,
                                                                      uint32_t B, uint32_t leafNum,
                                                                      kk_vdb_node *__restrict__ nodes) {
    __shared__ uint32_t s_count[LEAF_NODES_THREADS / 64];
    __shared__ uint8_t s_flag[LEAF_NODES_STAGED]; // the flags in memory order, when they fit
    const uint32_t total = lx * ly * lz, tid = threadIdx.x;
    const uint32_t per = (total + LEAF_NODES_THREADS - 1) / LEAF_NODES_THREADS;
    const uint32_t first = tid * per, last = first + per < total ? first + per : total;
    const bool staged = total <= LEAF_NODES_STAGED; // read once, coalesced; the ordered walks then read LDS
    if (staged) {
        for (uint32_t i = tid; i < total; i += LEAF_NODES_THREADS)
            s_flag[i] = leafValid[i] != 0u ? 1 : 0;
        __syncthreads();
    }
    // position n of the (x, y, z) order: x = n / (ly lz), y = (n / lz) % ly, z = n % lz; a walk keeps the
    // three coordinates and steps them (z fastest) instead of dividing at every position
    auto flagAt = [&](uint32_t x, uint32_t y, uint32_t z) {
        const size_t at = (static_cast<size_t>(z) * ly + y) * lx + x;
        return staged ? s_flag[at] != 0 : leafValid[at] != 0u;
    };
    auto step = [&](uint32_t &x, uint32_t &y, uint32_t &z) {
        if (++z == lz) {
            z = 0;
            if (++y == ly) {
                y = 0;
                ++x;
            }
        }
    };
    const uint32_t x0 = first / (ly * lz), y0 = (first / lz) % ly, z0 = first % lz;
    uint32_t mine = 0;
    {
        uint32_t x = x0, y = y0, z = z0;
        for (uint32_t n = first; n < last; ++n, step(x, y, z))
            mine += flagAt(x, y, z) ? 1u : 0u;
    }
    // exclusive scan of the threads' counts: inside the waves by shuffles, over the waves through LDS
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t incl = mine;
    for (uint32_t o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o)
            incl += t;
    }
    if (lane == 63)
        s_count[wave] = incl;
    __syncthreads();
    uint32_t k = incl - mine;
    for (uint32_t w = 0; w < wave; ++w)
        k += s_count[w];
    {
        uint32_t x = x0, y = y0, z = z0;
        for (uint32_t n = first; n < last; ++n, step(x, y, z))
            if (flagAt(x, y, z)) {
                if (k < leafNum) { // (more valid bricks than the plan counted: flags changed; the host checks)
                    nodes[k].pos[0] = static_cast<int32_t>(x * B);
                    nodes[k].pos[1] = static_cast<int32_t>(y * B);
                    nodes[k].pos[2] = static_cast<int32_t>(z * B);
                }
                ++k;
            }
    }
}

template <typename T>
int fillAtlas(const T *d_volume, uint32_t dx, uint32_t dy, uint32_t dz, const kk_vdb_plan *plan,
              const kk_vdb_node *d_node_pools, const T *h_rngs, uint32_t n, T *d_atlas, void *stream) {
    KK_REQUIRE(d_volume && plan && d_node_pools && d_atlas, "null pointer");
    KK_REQUIRE(plan->leaf_num > 0, "no leaves");
    Ranges<T> r;
    int rc = makeRanges(h_rngs, n, r);
    if (rc != KK_OK)
        return rc;
    const kk_vdb_param &p = plan->param;
    const size_t A = static_cast<size_t>(p.dims[0]) + 2u * p.apron_dep_wid;
    const size_t lds = 3 * A * A * sizeof(unsigned long long) + A * A * A * sizeof(T);
    if (A <= 64 && p.dims[0] >= 2 && lds <= 150 * 1024) {
        // the default dynamic LDS limit is 64 KB; the attribute belongs to the current device's copy
        // of the kernel, so it is set before every launch (a cheap host-side call)
        if (A == 36) {
            KK_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&fill_atlas_lds_kernel<T, 36>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            fill_atlas_lds_kernel<T, 36><<<plan->leaf_num, FILL_THREADS, lds, static_cast<hipStream_t>(stream)>>>(
                d_volume, dx, dy, dz, d_node_pools + p.node_starts[0], p.dims[0], p.apron_wid,
                p.apron_dep_wid, r, d_atlas);
        } else {
            KK_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&fill_atlas_lds_kernel<T, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            fill_atlas_lds_kernel<T, 0><<<plan->leaf_num, FILL_THREADS, lds, static_cast<hipStream_t>(stream)>>>(
                d_volume, dx, dy, dz, d_node_pools + p.node_starts[0], p.dims[0], p.apron_wid,
                p.apron_dep_wid, r, d_atlas);
        }
    } else
        fill_atlas_kernel<T><<<plan->leaf_num, THREADS, 0, static_cast<hipStream_t>(stream)>>>(
            d_volume, dx, dy, dz, d_node_pools + p.node_starts[0], p.dims[0], p.apron_wid,
            p.apron_dep_wid, r, d_atlas);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

// ---- host wiring -------------------------------------------------------------------------------------

struct Coord {
    int32_t x, y, z;
    bool operator<(const Coord &o) const {
        return x != o.x ? x < o.x : y != o.y ? y < o.y : z < o.z;
    }
    bool operator==(const Coord &o) const { return x == o.x && y == o.y && z == o.z; }
};

// Node coordinates of every level, each level sorted by (x, y, z) and unique: level 0 holds the valid
// leaves in brick units, level l+1 the level-l coordinates divided by dims[l+1].
static void levelCoords(const kk_vdb_plan &plan, const uint32_t *leafValid,
                        std::vector<Coord> levels[KK_VDB_MAX_LEVEL_NUM]) {
    const uint32_t lx = plan.leaf_slots[0], ly = plan.leaf_slots[1], lz = plan.leaf_slots[2];
    // x outermost: the leaves come out in (x, y, z) order, the order of the pools, without a sort
    for (uint32_t x = 0; x < lx; ++x)
        for (uint32_t y = 0; y < ly; ++y)
            for (uint32_t z = 0; z < lz; ++z)
                if (leafValid[(static_cast<size_t>(z) * ly + y) * lx + x])
                    levels[0].push_back({static_cast<int32_t>(x), static_cast<int32_t>(y),
                                         static_cast<int32_t>(z)});
    for (int lev = 1; lev <= plan.param.root_lev; ++lev) {
        const int32_t d = plan.param.dims[lev];
        for (const Coord &c : levels[lev - 1])
            levels[lev].push_back({c.x / d, c.y / d, c.z / d});
        std::sort(levels[lev].begin(), levels[lev].end());
        levels[lev].erase(std::unique(levels[lev].begin(), levels[lev].end()), levels[lev].end());
    }
}

} // namespace vdb
} // namespace kk

extern "C" int kk_vdb_classify_u8(const uint8_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                                  uint32_t log2_dim0, const uint8_t *h_empty_rngs, uint32_t n_rngs,
                                  uint32_t *d_leaf_valid, void *stream) {
    return kk::vdb::classify<uint8_t>(d_volume, dx, dy, dz, log2_dim0, h_empty_rngs, n_rngs,
                                      d_leaf_valid, stream);
}
extern "C" int kk_vdb_classify_u16(const uint16_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                                   uint32_t log2_dim0, const uint16_t *h_empty_rngs, uint32_t n_rngs,
                                   uint32_t *d_leaf_valid, void *stream) {
    return kk::vdb::classify<uint16_t>(d_volume, dx, dy, dz, log2_dim0, h_empty_rngs, n_rngs,
                                       d_leaf_valid, stream);
}

extern "C" int kk_vdb_plan_build(uint32_t dx, uint32_t dy, uint32_t dz, const uint8_t log2_dims[3],
                                 uint8_t apron_wid, int use_depth_box, const uint32_t *h_leaf_valid,
                                 kk_vdb_plan *out) {
    using namespace kk;
    KK_REQUIRE(log2_dims && h_leaf_valid && out, "null pointer");
    for (int i = 0; i < 3; ++i)
        KK_REQUIRE(log2_dims[i] > 0 && log2_dims[i] <= 10, "log2Dim > 0 && log2Dim <= MaxLog2Dim should be true");
    KK_REQUIRE(dx != 0 && dy != 0 && dz != 0, "volume extent should not be zero");
    memset(out, 0, sizeof(*out));
    kk_vdb_param &p = out->param;
    p.apron_wid = apron_wid;
    p.apron_dep_wid = apron_wid + (use_depth_box ? 1 : 0);
    for (int lev = 0; lev < KK_VDB_MAX_LEVEL_NUM; ++lev) {
        p.log2_dims[lev] = log2_dims[lev];
        p.dims[lev] = 1 << log2_dims[lev];
    }
    p.min_dep_pos_val_in_brick = -static_cast<int32_t>(p.apron_dep_wid);
    p.max_dep_pos_val_in_brick = p.dims[0] - 1 + p.apron_dep_wid;
    const uint32_t B = p.dims[0];
    out->leaf_slots[0] = (dx + B - 1) / B;
    out->leaf_slots[1] = (dy + B - 1) / B;
    out->leaf_slots[2] = (dz + B - 1) / B;

    // smallest level whose span covers the volume (dpbx_vdb.cu:301-338)
    uint64_t cover = B;
    p.voxs_per_child[0] = 1.f;
    int root = 0;
    while (cover < dx || cover < dy || cover < dz) {
        ++root;
        if (root == KK_VDB_MAX_LEVEL_NUM) {
            set_error("Current VDB configuration cannot cover %ux%ux%u voxels.", dx, dy, dz);
            return KK_ERR_INVALID;
        }
        p.voxs_per_child[root] = static_cast<float>(cover);
        cover *= static_cast<uint64_t>(p.dims[root]);
    }
    if (root == 0) {
        set_error("one brick covers the whole %ux%ux%u volume: the tree would have no inner level", dx,
                  dy, dz);
        return KK_ERR_INVALID;
    }
    p.root_lev = static_cast<uint8_t>(root);
    p.vox_per_vdb = static_cast<float>(cover);
    p.vox_per_vol[0] = static_cast<float>(dx);
    p.vox_per_vol[1] = static_cast<float>(dy);
    p.vox_per_vol[2] = static_cast<float>(dz);

    std::vector<vdb::Coord> levels[KK_VDB_MAX_LEVEL_NUM];
    vdb::levelCoords(*out, h_leaf_valid, levels);
    out->leaf_num = static_cast<uint32_t>(levels[0].size());
    if (out->leaf_num == 0) {
        set_error("every brick is empty under the given scalar ranges");
        return KK_ERR_INVALID;
    }
    p.vox_per_atlas_brick = p.dims[0] + 2 * p.apron_dep_wid;
    p.atlas_brick_per_vdb[0] = out->leaf_slots[0];
    p.atlas_brick_per_vdb[1] = out->leaf_slots[1];
    const uint64_t perLayer = static_cast<uint64_t>(out->leaf_slots[0]) * out->leaf_slots[1];
    p.atlas_brick_per_vdb[2] = static_cast<int32_t>((out->leaf_num + perLayer - 1) / perLayer);
    out->atlas_map_num = perLayer * p.atlas_brick_per_vdb[2];
    uint64_t nodes = 0, children = 0;
    for (int lev = 0; lev <= root; ++lev) {
        p.node_starts[lev] = nodes;
        nodes += levels[lev].size();
        if (lev > 0) {
            p.child_starts[lev - 1] = children;
            children += static_cast<uint64_t>(levels[lev].size()) * p.dims[lev] * p.dims[lev] * p.dims[lev];
        }
    }
    out->node_num = nodes;
    out->child_num = children;
    const uint64_t A = p.vox_per_atlas_brick;
    out->atlas_voxels = A * A * A * out->leaf_num;
    return KK_OK;
}

extern "C" int kk_vdb_leaf_nodes(const uint32_t *d_leaf_valid, const kk_vdb_plan *plan, kk_vdb_node *d_leaf_nodes,
                                 void *stream) {
    using namespace kk;
    KK_REQUIRE(d_leaf_valid && plan && d_leaf_nodes, "null pointer");
    KK_REQUIRE(plan->leaf_num > 0, "no leaves");
    const uint64_t total = static_cast<uint64_t>(plan->leaf_slots[0]) * plan->leaf_slots[1] * plan->leaf_slots[2];
    KK_REQUIRE(total > 0 && total < (1ull << 31), "brick count out of range");
    vdb::leaf_nodes_kernel<<<1, vdb::LEAF_NODES_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
        d_leaf_valid, plan->leaf_slots[0], plan->leaf_slots[1], plan->leaf_slots[2],
        static_cast<uint32_t>(plan->param.dims[0]), plan->leaf_num, d_leaf_nodes);
    KK_LAUNCH_CHECK();
    return KK_OK;
}

extern "C" int kk_vdb_wire(const kk_vdb_plan *plan, const uint32_t *h_leaf_valid,
                           kk_vdb_node *h_node_pools, uint32_t *h_child_pools,
                           uint32_t *h_atlas_brick_to_node) {
    using namespace kk;
    KK_REQUIRE(plan && h_leaf_valid && h_node_pools && h_child_pools && h_atlas_brick_to_node,
               "null pointer");
    const kk_vdb_param &p = plan->param;
    std::vector<vdb::Coord> levels[KK_VDB_MAX_LEVEL_NUM];
    vdb::levelCoords(*plan, h_leaf_valid, levels);
    KK_REQUIRE(levels[0].size() == plan->leaf_num, "validity flags changed since the plan was made");
    memset(h_node_pools, 0, sizeof(kk_vdb_node) * plan->node_num);
    for (uint64_t i = 0; i < plan->child_num; ++i)
        h_child_pools[i] = KK_VDB_UNDEF_RLT_IDX;
    for (uint64_t i = 0; i < plan->atlas_map_num; ++i)
        h_atlas_brick_to_node[i] = KK_VDB_UNDEF_RLT_IDX;

    const uint32_t lx = plan->leaf_slots[0];
    const uint32_t lYxX = lx * plan->leaf_slots[1];
    for (int lev = 0; lev < p.root_lev; ++lev) {
        // voxels spanned by one unit of this level's coordinates
        int64_t unit = 1;
        for (int l = 0; l <= lev; ++l)
            unit *= p.dims[l];
        const std::vector<vdb::Coord> &up = levels[lev + 1];
        const int L = p.log2_dims[lev + 1];
        const int32_t d = p.dims[lev + 1];
        const uint64_t childrenPerNode = static_cast<uint64_t>(d) * d * d;
        for (uint32_t k = 0; k < levels[lev].size(); ++k) {
            const vdb::Coord c = levels[lev][k];
            kk_vdb_node &n = h_node_pools[p.node_starts[lev] + k];
            n.pos[0] = static_cast<int32_t>(c.x * unit);
            n.pos[1] = static_cast<int32_t>(c.y * unit);
            n.pos[2] = static_cast<int32_t>(c.z * unit);
            if (lev == 0) {
                n.child_pool_offs_relative = KK_VDB_UNDEF_RLT_IDX;
                n.brick_pos_in_atlas[2] = k / lYxX;
                n.brick_pos_in_atlas[1] = (k % lYxX) / lx;
                n.brick_pos_in_atlas[0] = k % lx;
                h_atlas_brick_to_node[k] = k; // flat atlas index of brick k is k
            } else
                n.child_pool_offs_relative = static_cast<uint64_t>(k) * p.dims[lev] * p.dims[lev] * p.dims[lev];
            // hook the node into its parent's dense child block
            const vdb::Coord pc = {c.x / d, c.y / d, c.z / d};
            const uint64_t parent =
                (lev + 1 == p.root_lev)
                    ? 0
                    : static_cast<uint64_t>(std::lower_bound(up.begin(), up.end(), pc) - up.begin());
            const uint32_t rx = c.x - pc.x * d, ry = c.y - pc.y * d, rz = c.z - pc.z * d;
            h_child_pools[p.child_starts[lev] + parent * childrenPerNode +
                          ((rz << (2 * L)) | (ry << L) | rx)] = k;
        }
    }
    // the root sits last, at the origin, and owns the first child block of its level
    kk_vdb_node &rootNode = h_node_pools[plan->node_num - 1];
    memset(&rootNode, 0, sizeof(rootNode));
    return KK_OK;
}

extern "C" int kk_vdb_fill_atlas_u8(const uint8_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                                    const kk_vdb_plan *plan, const kk_vdb_node *d_node_pools,
                                    const uint8_t *h_empty_rngs, uint32_t n_rngs, uint8_t *d_atlas,
                                    void *stream) {
    return kk::vdb::fillAtlas<uint8_t>(d_volume, dx, dy, dz, plan, d_node_pools, h_empty_rngs, n_rngs,
                                       d_atlas, stream);
}
extern "C" int kk_vdb_fill_atlas_u16(const uint16_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                                     const kk_vdb_plan *plan, const kk_vdb_node *d_node_pools,
                                     const uint16_t *h_empty_rngs, uint32_t n_rngs, uint16_t *d_atlas,
                                     void *stream) {
    return kk::vdb::fillAtlas<uint16_t>(d_volume, dx, dy, dz, plan, d_node_pools, h_empty_rngs, n_rngs,
                                        d_atlas, stream);
}
