// Multi-GPU plumbing of the C ABI: one process per GPU, a structure built once on one rank is
// broadcast to the others and per-rank image stripes are gathered to one rank, over RCCL (xGMI inside
// a node). The reference renders on one GPU and has no counterpart.
//
// RCCL is not linked: it is opened at the first kk_comm_* call (the copy a host process has already
// loaded - PyTorch brings its own - is reused, else the system's), so that hosts without a second GPU
// never need it and two copies never meet in one process.
//
// Layout for broadcasting: a structure lives in ONE device allocation ("blob") whose pieces sit at
// fixed offsets computed from sizes every rank knows, so a broadcast is a single collective over the
// blob's bytes; device pointers inside it are rewritten locally afterwards (kk_lbvh_blob_bind,
// kk_vdb_blob_device_data).
#include "kk_host.h"

#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include <string.h>

struct kk_comm {
    ncclComm_t comm;
    int rank, nranks, device;
    void *d_stage; // 4 KB staging buffer of kk_comm_broadcast_host
};

namespace kk {
namespace comm {

struct Api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Api g_api;
static std::mutex g_apiMutex;

static int load_api() {
    std::lock_guard<std::mutex> lock(g_apiMutex);
    if (g_api.handle)
        return KK_OK;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) // a copy the process already holds
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr)
            break;
    for (int i = 0; !h && i < 4; ++i)
        h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        set_error("RCCL cannot be loaded: %s", dlerror());
        return KK_ERR_HIP;
    }
    Api a;
    a.handle = h;
#define KK_SYM(field, name)                                                                        \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                                 \
    if (!a.field) {                                                                                \
        set_error("RCCL symbol %s is missing", name);                                              \
        return KK_ERR_HIP;                                                                         \
    }
    KK_SYM(GetUniqueId, "ncclGetUniqueId")
    KK_SYM(CommInitRank, "ncclCommInitRank")
    KK_SYM(CommDestroy, "ncclCommDestroy")
    KK_SYM(Broadcast, "ncclBroadcast")
    KK_SYM(Send, "ncclSend")
    KK_SYM(Recv, "ncclRecv")
    KK_SYM(GroupStart, "ncclGroupStart")
    KK_SYM(GroupEnd, "ncclGroupEnd")
    KK_SYM(GetErrorString, "ncclGetErrorString")
#undef KK_SYM
    g_api = a;
    return KK_OK;
}

#define KK_NCCL_TRY(call)                                                                          \
    do {                                                                                           \
        ncclResult_t kk_r_ = (call);                                                               \
        if (kk_r_ != ncclSuccess) {                                                                \
            kk::set_error("%s failed at %s:%d: %s", #call, __FILE__, __LINE__,                     \
                          kk::comm::g_api.GetErrorString(kk_r_));                                  \
            return KK_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

constexpr int STRIPE_ROWS = 16;

// frame row y <- block (stripe % nranks), slot (stripe / nranks), row y % 16, stripe = y / 16
__global__ __launch_bounds__(256) void unstripe_kernel(const uint4 *__restrict__ blocks, int nranks, int per,
                                                       int rowQuads, int height, uint4 *__restrict__ rgba) {
    const int y = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= rowQuads || y >= height)
        return;
    const int stripe = y / STRIPE_ROWS;
    const size_t src = ((static_cast<size_t>(stripe % nranks) * per + stripe / nranks) * STRIPE_ROWS +
                        y % STRIPE_ROWS) * rowQuads + q;
    rgba[static_cast<size_t>(y) * rowQuads + q] = blocks[src];
}
__global__ __launch_bounds__(256) void unstripe_pixels_kernel(const uint32_t *__restrict__ blocks, int nranks,
                                                              int per, int width, int height,
                                                              uint32_t *__restrict__ rgba) {
    const int y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= width || y >= height)
        return;
    const int stripe = y / STRIPE_ROWS;
    const size_t src = ((static_cast<size_t>(stripe % nranks) * per + stripe / nranks) * STRIPE_ROWS +
                        y % STRIPE_ROWS) * width + x;
    rgba[static_cast<size_t>(y) * width + x] = blocks[src];
}

struct LbvhBlob {
    size_t deviceData, positions, facePosIdx, faces, compacted, nodes, total;
};
inline LbvhBlob lbvh_blob(uint32_t N, uint32_t V) {
    LbvhBlob l;
    Carver c;
    l.deviceData = c.take(sizeof(kk_lbvh_device_data));
    l.positions = c.take(12ull * V);
    l.facePosIdx = c.take(12ull * N);
    l.faces = c.take(4ull * N);
    l.compacted = c.take(4ull * N);
    l.nodes = c.take(sizeof(kk_lbvh_node) * static_cast<size_t>(N));
    l.total = c.off;
    return l;
}

__global__ void lbvh_bind_kernel(char *blob, LbvhBlob l) {
    auto *dd = reinterpret_cast<kk_lbvh_device_data *>(blob + l.deviceData);
    dd->faces = reinterpret_cast<uint32_t *>(blob + l.faces);
    dd->compacted_faces = reinterpret_cast<uint32_t *>(blob + l.compacted);
    dd->internal_nodes = reinterpret_cast<kk_lbvh_node *>(blob + l.nodes);
}

struct SceneBlob {
    size_t triangles, triToPositionIndices, positions, normals, texCoords, lights, materials, total;
};
inline SceneBlob scene_blob(const kk_pt_scene_sizes &z) {
    SceneBlob l;
    Carver c;
    l.triangles = c.take(sizeof(kk_pt_triangle) * static_cast<size_t>(z.face_num));
    l.triToPositionIndices = c.take(12ull * z.face_num);
    l.positions = c.take(12ull * z.vert_num);
    l.normals = c.take(12ull * z.normal_num);
    l.texCoords = c.take(8ull * z.tex_coord_num);
    l.lights = c.take(sizeof(kk_light) * static_cast<size_t>(z.light_num));
    l.materials = c.take(sizeof(kk_material) * static_cast<size_t>(z.material_num));
    l.total = c.off;
    return l;
}
inline bool scene_sizes_ok(const kk_pt_scene_sizes &z) {
    return z.face_num > 0 && z.face_num <= KK_LBVH_MAX_FACES && z.vert_num > 0 && z.normal_num > 0 &&
           z.material_num > 0;
}

struct VdbBlob {
    size_t nodes, children, atlasMap, atlas, total;
};
inline VdbBlob vdb_blob(const kk_vdb_plan &p, uint32_t voxelBytes) {
    VdbBlob l;
    Carver c;
    l.nodes = c.take(sizeof(kk_vdb_node) * static_cast<size_t>(p.node_num));
    l.children = c.take(4ull * p.child_num);
    l.atlasMap = c.take(4ull * p.atlas_map_num);
    l.atlas = c.take(static_cast<size_t>(p.atlas_voxels) * voxelBytes);
    l.total = c.off;
    return l;
}

} // namespace comm
} // namespace kk

using namespace kk;
using namespace kk::comm;

extern "C" int kk_comm_unique_id(void *h_id) {
    KK_REQUIRE(h_id != nullptr, "null pointer");
    static_assert(sizeof(ncclUniqueId) == KK_COMM_ID_BYTES, "id size");
    const int rc = load_api();
    if (rc != KK_OK)
        return rc;
    ncclUniqueId id;
    KK_NCCL_TRY(g_api.GetUniqueId(&id));
    memcpy(h_id, &id, sizeof(id));
    return KK_OK;
}

extern "C" int kk_comm_init(int nranks, int rank, const void *h_id, kk_comm **out) {
    KK_REQUIRE(h_id && out, "null pointer");
    KK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank out of range");
    const int rc = load_api();
    if (rc != KK_OK)
        return rc;
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    kk_comm *c = new kk_comm{};
    c->rank = rank;
    c->nranks = nranks;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        set_error("hipGetDevice failed");
        return KK_ERR_HIP;
    }
    ncclResult_t r = g_api.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", g_api.GetErrorString(r));
        delete c;
        return KK_ERR_HIP;
    }
    if (hipMalloc(&c->d_stage, 4096) != hipSuccess) {
        g_api.CommDestroy(c->comm);
        delete c;
        set_error("hipMalloc of the staging buffer failed");
        return KK_ERR_HIP;
    }
    *out = c;
    return KK_OK;
}

// The communicator belongs to the device that was current at kk_comm_init: its collectives run on that
// device's streams and its staging buffer lives there.
static int on_own_device(const kk_comm *c) {
    int dev = -1;
    KK_HIP_TRY(hipGetDevice(&dev));
    if (dev != c->device) {
        set_error("the communicator was made on device %d, the current device is %d", c->device, dev);
        return KK_ERR_INVALID;
    }
    return KK_OK;
}

extern "C" int kk_comm_destroy(kk_comm *c) {
    if (!c)
        return KK_OK;
    // collectives issued on any stream of the communicator's device may still be in flight (a gather on
    // a side stream): they finish before the communicator and its staging buffer go away
    int prev = -1;
    const bool switched = hipGetDevice(&prev) == hipSuccess && prev != c->device &&
                          hipSetDevice(c->device) == hipSuccess;
    (void)hipDeviceSynchronize();
    if (c->d_stage)
        (void)hipFree(c->d_stage);
    ncclResult_t r = g_api.CommDestroy(c->comm);
    if (switched)
        (void)hipSetDevice(prev);
    delete c;
    if (r != ncclSuccess) {
        set_error("ncclCommDestroy failed: %s", g_api.GetErrorString(r));
        return KK_ERR_HIP;
    }
    return KK_OK;
}

extern "C" int kk_comm_rank(const kk_comm *c, int *rank, int *nranks) {
    KK_REQUIRE(c != nullptr, "null pointer");
    if (rank)
        *rank = c->rank;
    if (nranks)
        *nranks = c->nranks;
    return KK_OK;
}

extern "C" int kk_comm_broadcast(kk_comm *c, void *d_buf, size_t bytes, int root, void *stream) {
    KK_REQUIRE(c && d_buf, "null pointer");
    KK_REQUIRE(root >= 0 && root < c->nranks, "root out of range");
    if (bytes == 0)
        return KK_OK;
    if (const int rc = on_own_device(c); rc != KK_OK)
        return rc;
    KK_NCCL_TRY(g_api.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, c->comm, static_cast<hipStream_t>(stream)));
    return KK_OK;
}

extern "C" int kk_comm_broadcast_host(kk_comm *c, void *h_buf, size_t bytes, int root, void *stream_) {
    KK_REQUIRE(c && h_buf, "null pointer");
    KK_REQUIRE(bytes <= 4096, "at most 4096 bytes");
    KK_REQUIRE(root >= 0 && root < c->nranks, "root out of range");
    if (const int rc = on_own_device(c); rc != KK_OK)
        return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (c->rank == root)
        KK_HIP_TRY(hipMemcpyAsync(c->d_stage, h_buf, bytes, hipMemcpyHostToDevice, stream));
    KK_NCCL_TRY(g_api.Broadcast(c->d_stage, c->d_stage, bytes, ncclUint8, root, c->comm, stream));
    KK_HIP_TRY(hipMemcpyAsync(h_buf, c->d_stage, bytes, hipMemcpyDeviceToHost, stream));
    KK_HIP_TRY(hipStreamSynchronize(stream));
    return KK_OK;
}

extern "C" int kk_comm_gather(kk_comm *c, const void *d_send, void *d_recv, size_t bytes_per_rank, int root,
                              void *stream_) {
    KK_REQUIRE(c && d_send, "null pointer");
    KK_REQUIRE(root >= 0 && root < c->nranks, "root out of range");
    KK_REQUIRE(c->rank != root || d_recv != nullptr, "the root needs a receive buffer");
    if (bytes_per_rank == 0)
        return KK_OK;
    if (const int rc = on_own_device(c); rc != KK_OK)
        return rc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (c->rank == root) {
        // the root's own block is a local copy; the others arrive over one link each
        KK_HIP_TRY(hipMemcpyAsync(static_cast<char *>(d_recv) + bytes_per_rank * root, d_send, bytes_per_rank,
                                  hipMemcpyDeviceToDevice, stream));
        if (c->nranks > 1) {
            KK_NCCL_TRY(g_api.GroupStart());
            for (int r = 0; r < c->nranks; ++r)
                if (r != root)
                    KK_NCCL_TRY(g_api.Recv(static_cast<char *>(d_recv) + bytes_per_rank * r, bytes_per_rank,
                                           ncclUint8, r, c->comm, stream));
            KK_NCCL_TRY(g_api.GroupEnd());
        }
    } else
        KK_NCCL_TRY(g_api.Send(d_send, bytes_per_rank, ncclUint8, root, c->comm, stream));
    return KK_OK;
}

// ---- LBVH blob ---------------------------------------------------------------------------------------

extern "C" int kk_lbvh_blob_layout_of(uint32_t face_num, uint32_t vert_num, kk_lbvh_blob_layout *out) {
    KK_REQUIRE(out != nullptr, "null pointer");
    KK_REQUIRE(face_num > 0 && face_num <= KK_LBVH_MAX_FACES && vert_num > 0, "sizes out of range");
    const LbvhBlob l = lbvh_blob(face_num, vert_num);
    out->device_data = l.deviceData;
    out->positions = l.positions;
    out->face_pos_idx = l.facePosIdx;
    out->faces = l.faces;
    out->compacted = l.compacted;
    out->nodes = l.nodes;
    out->total = l.total;
    return KK_OK;
}

extern "C" int kk_lbvh_blob_bind(void *d_blob, uint32_t face_num, uint32_t vert_num, void *stream) {
    KK_REQUIRE(d_blob != nullptr, "null pointer");
    KK_REQUIRE((reinterpret_cast<uintptr_t>(d_blob) & 255u) == 0, "d_blob must be 256-byte aligned");
    KK_REQUIRE(face_num > 0 && face_num <= KK_LBVH_MAX_FACES && vert_num > 0, "sizes out of range");
    lbvh_bind_kernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(static_cast<char *>(d_blob),
                                                                  lbvh_blob(face_num, vert_num));
    KK_LAUNCH_CHECK();
    return KK_OK;
}

extern "C" int kk_lbvh_broadcast(kk_comm *c, void *d_blob, uint32_t face_num, uint32_t vert_num, int root,
                                 void *stream) {
    KK_REQUIRE(c && d_blob, "null pointer");
    KK_REQUIRE(face_num > 0 && face_num <= KK_LBVH_MAX_FACES && vert_num > 0, "sizes out of range");
    int rc = kk_comm_broadcast(c, d_blob, lbvh_blob(face_num, vert_num).total, root, stream);
    if (rc != KK_OK)
        return rc;
    return kk_lbvh_blob_bind(d_blob, face_num, vert_num, stream);
}

// ---- path tracer's scene blob ------------------------------------------------------------------------

extern "C" int kk_pt_scene_blob_layout_of(const kk_pt_scene_sizes *sizes, kk_pt_scene_blob_layout *out) {
    KK_REQUIRE(sizes && out, "null pointer");
    KK_REQUIRE(scene_sizes_ok(*sizes), "sizes out of range");
    const SceneBlob l = scene_blob(*sizes);
    out->triangles = l.triangles;
    out->tri_to_position_indices = l.triToPositionIndices;
    out->positions = l.positions;
    out->normals = l.normals;
    out->tex_coords = l.texCoords;
    out->lights = l.lights;
    out->materials = l.materials;
    out->total = l.total;
    return KK_OK;
}

extern "C" int kk_pt_scene_blob_bind(void *d_blob, const kk_pt_scene_sizes *sizes, kk_pt_scene *h_io) {
    KK_REQUIRE(d_blob && sizes && h_io, "null pointer");
    KK_REQUIRE(scene_sizes_ok(*sizes), "sizes out of range");
    const SceneBlob l = scene_blob(*sizes);
    const char *b = static_cast<const char *>(d_blob);
    h_io->triangles = reinterpret_cast<const kk_pt_triangle *>(b + l.triangles);
    h_io->tri_to_position_indices = reinterpret_cast<const uint32_t *>(b + l.triToPositionIndices);
    h_io->positions = reinterpret_cast<const float *>(b + l.positions);
    h_io->normals = reinterpret_cast<const float *>(b + l.normals);
    h_io->tex_coords = sizes->tex_coord_num ? reinterpret_cast<const float *>(b + l.texCoords) : nullptr;
    h_io->lights = sizes->light_num ? reinterpret_cast<const kk_light *>(b + l.lights) : nullptr;
    h_io->materials = reinterpret_cast<const kk_material *>(b + l.materials);
    h_io->light_num = sizes->light_num;
    h_io->light_tot_area = sizes->light_tot_area;
    return KK_OK;
}

extern "C" int kk_pt_scene_broadcast(kk_comm *c, void *d_blob, const kk_pt_scene_sizes *sizes, int root,
                                     void *stream) {
    KK_REQUIRE(c && d_blob && sizes, "null pointer");
    KK_REQUIRE(scene_sizes_ok(*sizes), "sizes out of range");
    return kk_comm_broadcast(c, d_blob, scene_blob(*sizes).total, root, stream);
}

// ---- VDB blob ----------------------------------------------------------------------------------------

extern "C" int kk_vdb_blob_layout_of(const kk_vdb_plan *plan, uint32_t voxel_bytes, kk_vdb_blob_layout *out) {
    KK_REQUIRE(plan && out, "null pointer");
    KK_REQUIRE(voxel_bytes == 1 || voxel_bytes == 2, "voxel_bytes is 1 or 2");
    const VdbBlob l = vdb_blob(*plan, voxel_bytes);
    out->node_pools = l.nodes;
    out->child_pools = l.children;
    out->atlas_map = l.atlasMap;
    out->atlas = l.atlas;
    out->total = l.total;
    return KK_OK;
}

extern "C" int kk_vdb_blob_device_data(void *d_blob, const kk_vdb_plan *plan, uint32_t voxel_bytes,
                                       kk_vdb_device_data *h_out) {
    KK_REQUIRE(d_blob && plan && h_out, "null pointer");
    KK_REQUIRE(voxel_bytes == 1 || voxel_bytes == 2, "voxel_bytes is 1 or 2");
    const VdbBlob l = vdb_blob(*plan, voxel_bytes);
    char *b = static_cast<char *>(d_blob);
    h_out->node_pools = reinterpret_cast<kk_vdb_node *>(b + l.nodes);
    h_out->child_pools = reinterpret_cast<uint32_t *>(b + l.children);
    h_out->atlas_brick_to_node_indices = reinterpret_cast<uint32_t *>(b + l.atlasMap);
    h_out->atlas = b + l.atlas;
    h_out->leaf_num = plan->leaf_num;
    h_out->voxel_bytes = voxel_bytes;
    h_out->vdb_param = plan->param;
    return KK_OK;
}

extern "C" int kk_vdb_broadcast(kk_comm *c, void *d_blob, const kk_vdb_plan *plan, uint32_t voxel_bytes,
                                int root, void *stream) {
    KK_REQUIRE(c && d_blob && plan, "null pointer");
    KK_REQUIRE(voxel_bytes == 1 || voxel_bytes == 2, "voxel_bytes is 1 or 2");
    return kk_comm_broadcast(c, d_blob, vdb_blob(*plan, voxel_bytes).total, root, stream);
}

// ---- image stripes -----------------------------------------------------------------------------------

extern "C" int kk_stripes_per_rank(int height, int nranks) {
    if (height <= 0 || nranks <= 0)
        return 0;
    const int stripes = (height + STRIPE_ROWS - 1) / STRIPE_ROWS;
    return (stripes + nranks - 1) / nranks;
}

extern "C" int kk_gather_stripes(kk_comm *c, const uint8_t *d_stripes, uint8_t *d_blocks, int width, int height,
                                 int root, void *stream) {
    KK_REQUIRE(c != nullptr, "null pointer");
    KK_REQUIRE(width > 0 && height > 0, "bad image size");
    const size_t bytes = static_cast<size_t>(kk_stripes_per_rank(height, c->nranks)) * STRIPE_ROWS * width * 4;
    return kk_comm_gather(c, d_stripes, d_blocks, bytes, root, stream);
}

extern "C" int kk_unstripe_rgba(const uint8_t *d_blocks, int nranks, int width, int height, uint8_t *d_rgba,
                                void *stream_) {
    KK_REQUIRE(d_blocks && d_rgba, "null pointer");
    KK_REQUIRE(width > 0 && height > 0 && nranks > 0, "bad image size or rank count");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int per = kk_stripes_per_rank(height, nranks);
    const bool quads = width % 4 == 0 && (reinterpret_cast<uintptr_t>(d_blocks) & 15u) == 0 &&
                       (reinterpret_cast<uintptr_t>(d_rgba) & 15u) == 0;
    if (quads) {
        const int rowQuads = width / 4;
        unstripe_kernel<<<dim3(div_up(rowQuads, 256), height), 256, 0, stream>>>(
            reinterpret_cast<const uint4 *>(d_blocks), nranks, per, rowQuads, height,
            reinterpret_cast<uint4 *>(d_rgba));
    } else
        unstripe_pixels_kernel<<<dim3(div_up(width, 256), height), 256, 0, stream>>>(
            reinterpret_cast<const uint32_t *>(d_blocks), nranks, per, width, height,
            reinterpret_cast<uint32_t *>(d_rgba));
    KK_LAUNCH_CHECK();
    return KK_OK;
}
