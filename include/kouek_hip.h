/* C ABI of the MI355X (gfx950) implementation of the two data-parallel paths of Kouek/RayTracer:
 *   path 1: LBVH build, BVH traversal / intersection, path-tracing integrator
 *   path 2: Depth-Box VDB build, direct-volume-rendering ray marcher
 *
 * The reference has no FFI layer: its boundary is the public C++ API of four classes
 * (include/ray_tracer/lbvh.h, include/ray_tracer/ray_tracer.h, include/ray_caster/dpbx_vdb.h,
 * include/ray_caster/ray_caster.h). Each entry point below names the reference member function whose
 * GPU work it performs; the C++ classes in include/kouek/ keep the reference's signatures and call
 * these.
 *
 * Conventions
 *   - every function returns 0 on success and a negative kk_status otherwise; kk_last_error() gives
 *     the message of the calling thread's last failure (the reference prints and leaves
 *     IsComplete() == false, src/ray_caster/dpbx_vdb.cu:17-22; the C++ wrappers do the same);
 *   - pointers named d_* are device pointers, h_* host pointers; nothing here takes ownership;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream). Launchers enqueue work
 *     and return; they neither allocate nor synchronise unless their comment says so;
 *   - structs are plain C with the reference's memory layout where the reference has a counterpart.
 */
#ifndef KOUEK_HIP_H
#define KOUEK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum kk_status {
    KK_OK = 0,
    KK_ERR_INVALID = -1,   /* bad argument (null pointer, zero size, size over a limit) */
    KK_ERR_WORKSPACE = -2, /* workspace too small */
    KK_ERR_HIP = -3,       /* a HIP runtime call failed; see kk_last_error() */
    KK_ERR_DEVICE = -4     /* the device reported a failure flag (e.g. a look-back spin ran out) */
} kk_status;

const char *kk_last_error(void);
/* Compiled architecture and library version, e.g. "gfx950;0.1". */
const char *kk_build_info(void);

/* ---- plumbing for clients that own no device allocator (the C++ classes, ctypes tests) --------- */
int kk_set_device(int device);
int kk_malloc(void **d_ptr, size_t bytes);
int kk_free(void *d_ptr);
int kk_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int kk_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int kk_memset(void *d_dst, int value, size_t bytes, void *stream);
int kk_stream_synchronize(void *stream);
/* A HIP stream of the caller's own (hipStreamNonBlocking), for work that should overlap: every call
 * carries its parameters by value, so two DVR frames (or two ray batches) queued on two streams with
 * their own outputs and workspaces run side by side and fill each other's idle tails (4K DVR: 1391
 * frames/s against 949 one after the other). No counterpart in the reference, which renders one frame
 * at a time on the default stream. */
/* For a mesh that already lives on the device: KK_ERR_INVALID when some d_indices[i] >= limit (the kernels
 * follow face indices into positions / normals / texture coordinates / materials without a check of their
 * own, and one past an array is a GPU memory fault). d_word: four bytes of device memory for the result;
 * waits for `stream`. Not part of the reference, which uploads a mesh as its loader left it; the C++ and
 * Python host classes make the same check on the host before they upload. */
int kk_check_indices(const uint32_t *d_indices, size_t count, uint32_t limit, uint32_t *d_word, void *stream);
int kk_stream_create(void **out_stream);
int kk_stream_destroy(void *stream);

/* ================================================================================================
 * Path 1a: LBVH build  --  kouek::RayTracer::LBVH::BuildFrom (src/ray_tracer/lbvh.cu:100-323)
 * ============================================================================================== */

/* include/ray_tracer/lbvh.h:24-34. MSB of lft/rht set = the child is a leaf (compact index). */
typedef struct kk_lbvh_node {
    uint32_t lft, rht;
    float aabb_min[3], aabb_max[3];
} kk_lbvh_node; /* 32 bytes */

/* include/ray_tracer/lbvh.h:35-42, same offsets; lives in device memory, mirrored to the host. */
typedef struct kk_lbvh_device_data {
    float root_aabb_min[3], root_aabb_max[3];
    uint32_t face_num;
    uint32_t compacted_face_num;
    uint32_t *faces;           /* [face_num] sorted face permutation */
    uint32_t *compacted_faces; /* [compacted_face_num] start of each run of equal Morton codes */
    kk_lbvh_node *internal_nodes; /* [max(compacted_face_num - 1, 1)] */
    uint64_t _pad;
} kk_lbvh_device_data; /* 64 bytes */

#define KK_LBVH_MAX_FACES 0x3FFFFFFFu /* look-back words keep 30 bits of count */

enum {
    KK_LBVH_KEEP_STAGES = 1 /* also keep the unsorted Morton codes (one extra N*4 copy) for stage checks */
};

/* Byte offsets of the intermediate arrays inside the workspace, for stage-by-stage parity checks.
 * Valid after kk_lbvh_build has completed on the stream. */
typedef struct kk_lbvh_ws_layout {
    size_t total_bytes;
    size_t sorted_mortons;   /* uint32[N]  Morton codes in sorted order (lbvh.cu:172) */
    size_t unsorted_mortons; /* uint32[N]  only with KK_LBVH_KEEP_STAGES (lbvh.cu:136-170) */
    size_t unique_mortons;   /* uint32[M]  one code per leaf (lbvh.cu:202) */
    size_t leaf_aabbs;       /* float[M][6] per-leaf boxes (lbvh.cu:204-216) */
    size_t node_parents;     /* uint32[M-1] parent of each internal node (node 0: 0xFFFFFFFF); KEEP_STAGES only */
    size_t leaf_parents;     /* uint32[M]  parent of each leaf; KEEP_STAGES only */
    size_t error_flag;       /* uint32     non-zero if a sort look-back gave up */
} kk_lbvh_ws_layout;

int kk_lbvh_workspace_layout(uint32_t face_num, uint32_t vert_num, uint32_t flags,
                             kk_lbvh_ws_layout *out);

/* Builds the tree on `stream`: root bounds over all vert_num positions, 30-bit Morton code of each
 * triangle's box centre, stable radix sort, runs of equal codes -> leaves, Karras links, bottom-up
 * box refit. No host synchronisation: M (compacted_face_num) stays on the device until the caller
 * reads *d_out back.
 *   d_positions      float[vert_num][3]
 *   d_face_pos_idx   uint32[face_num][3]
 *   d_faces          uint32[face_num]        out
 *   d_compacted      uint32[face_num]        out (M used)
 *   d_nodes          kk_lbvh_node[face_num]  out (max(M-1,1) used), 16-byte aligned
 *   d_out            kk_lbvh_device_data     out; its pointers are set to the three arrays above
 * When M == 1 the reference allocates no internal node yet its traversal reads node 0; here node 0 is
 * then written with both children = leaf 0 and the leaf's box. face_num == 0 is rejected. */
int kk_lbvh_build(const float *d_positions, uint32_t vert_num, const uint32_t *d_face_pos_idx,
                  uint32_t face_num, uint32_t *d_faces, uint32_t *d_compacted,
                  kk_lbvh_node *d_nodes, kk_lbvh_device_data *d_out, void *d_workspace,
                  size_t workspace_bytes, uint32_t flags, void *stream);

/* The sort on its own (stable, ascending, 32-bit keys with 32-bit payloads; d_vals_in == NULL means
 * payload = index), as lbvh.cu:110,172 uses thrust::sequence + sort_by_key. Result lands in
 * d_keys_a / d_vals_a; the *_b arrays and d_workspace (kk_sort_workspace_bytes) are scratch. */
size_t kk_sort_workspace_bytes(uint32_t n);
/* Byte offset inside d_workspace of a uint32 that is non-zero after the sort when a tile's look-back
 * gave up waiting for its predecessor (the output is then not sorted): read it back after
 * synchronising when the result must be trusted. d_keys_a must be 16-byte aligned. */
size_t kk_sort_workspace_error_offset(uint32_t n);
int kk_sort_pairs_u32(uint32_t *d_keys_a, uint32_t *d_keys_b, uint32_t *d_vals_a, uint32_t *d_vals_b,
                      int vals_are_iota, uint32_t n, void *d_workspace, size_t workspace_bytes,
                      void *stream);
/* Waits for `stream` and reads that word back: KK_ERR_DEVICE when the last sort of n keys with this
 * workspace gave up (thrust::sort_by_key cannot fail this way; this sort's look-back can, see below). */
int kk_sort_workspace_status(const void *d_workspace, uint32_t n, void *stream);
/* How a tile of the sort (and of the LBVH build's run-head pass) finds its place in the chain of tiles.
 * Off (the default): a tile is its workgroup's index. That leans on workgroups being started in index
 * order, which holds for a grid that runs alone or beside kernels that do not wait on one another; two
 * look-back grids that fill the GPU at the same time (sorts or builds on several streams) can in
 * principle starve each other's tiles in front, which ends in the spin limit and the error word above.
 * On: a tile takes a ticket from an atomic counter instead (about 12 us more per pass of 10 M keys),
 * which holds whatever else runs. Callers that run sorts, LBVH builds, sorted closest-hit batches or
 * tile-order sorts concurrently on one GPU should switch it on. Process-wide; returns the old value. */
int kk_sort_set_ticketed(int on);
/* Test hook: while on, every sort raises its error word when it is done, as a look-back that gave up
 * would (the arrays it wrote are still sorted). Returns the old value. */
int kk_debug_sort_force_error(int on);

/* The same sort for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc). No counterpart in the reference. It takes no
 * workspace: the histograms, tile counters and look-back rows live in a device array that is part of
 * the library image, sized for n <= KK_SORT_STATIC_MAX_N (about 4.2 MB, held by every GPU the library
 * runs on until the process ends). Larger n is rejected with KK_ERR_INVALID. The call allocates
 * nothing and frees nothing; it clears the part of the array it uses on `stream` first. All calls on
 * one GPU share that array: calls that overlap in time on one GPU (different streams or threads) race.
 * Result as kk_sort_pairs_u32: in d_keys_a / d_vals_a. d_keys_a must be 16-byte aligned. */
#define KK_SORT_STATIC_MAX_N (1u << 22)
int kk_sort_pairs_u32_static(uint32_t *d_keys_a, uint32_t *d_keys_b, uint32_t *d_vals_a,
                             uint32_t *d_vals_b, int vals_are_iota, uint32_t n, void *stream);
/* Looks up the current device's copy of that array (hipGetSymbolAddress, which may load the code
 * object). kk_sort_pairs_u32_static does this by itself on its first call per device; a caller that
 * will make that first call under HIP graph capture calls this once per device beforehand instead. */
int kk_sort_static_init(void);
/* The static sort's error word (see kk_sort_workspace_status) after the work queued on `stream`. */
int kk_sort_static_status(void *stream);

/* ================================================================================================
 * Path 1b: closest hit  --  rayCastLBVH<false> (src/ray_tracer/ray_tracer_render.cu:192-270)
 * ============================================================================================== */

enum {
    KK_TRAVERSE_ORDERED = 0 /* near child first, boxes beyond the best hit skipped; ties in t go to the
                               smaller position in `faces`, which is the hit the reference's fixed
                               left-to-right walk keeps */
};

/* d_rays float[R][6] (pos, dir) -> d_t float[R] (FLT_MAX on a miss), d_bary float[R][2],
 * d_prim uint32[R] (face index, 0xFFFFFFFF on a miss). d_counters (optional, uint64[4], zeroed by the
 * caller) accumulates node boxes fetched, triangles tested, rays traced. d_workspace holds the ray
 * queue head (kk_closest_hit_workspace_bytes()). internal_nodes must be 16-byte aligned, as the
 * reference's KOUEK_CUDA_ALIGN makes them. Same slab and triangle arithmetic as
 * include/cg/ray.h:29-58.
 * A ray with a NaN or an infinite component reports a miss (here and in kk_closest_hit_accel and the
 * path tracer's walks). The reference's slab test calls every box "not missed" for such a ray, so it
 * walks the whole tree and keeps nothing (no NaN distance is the smaller one; only an infinite direction
 * can leave it a spurious hit at t = 0): the same answer without a visit of every node per bad ray. */
size_t kk_closest_hit_workspace_bytes(void);
int kk_closest_hit(const kk_lbvh_device_data *d_lbvh, const float *d_positions,
                   const uint32_t *d_face_pos_idx, const float *d_rays, uint32_t ray_num, float *d_t,
                   float *d_bary, uint32_t *d_prim, int mode, unsigned long long *d_counters,
                   void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- trace-side layout of a built tree (no counterpart in the reference: its rayCastLBVH walks the
 * build's own arrays, src/ray_tracer/ray_tracer_render.cu:192-270). kk_trace_accel_build repacks the
 * binary tree kk_lbvh_build produced into 8-wide nodes: a 128-byte record (one L2 line) holds up to
 * eight descendants of a binary node (the child with the largest box is opened until eight slots
 * are filled) as 8-bit quantised, outward-rounded boxes; a slot is a subtree (its record sits at
 * childBase + rank: children are contiguous) or a leaf (a leaf's box is the box of its triangles
 * grown by 1e-5 of the scene's largest coordinate). Triangles are 48-byte records (three vertices,
 * face index, the leaf's parent binary node, sorted position) stored contiguously per wide node.
 * Each binary node's parent index is kept for rays with a zero direction component.
 * kk_closest_hit_accel gives the same (t, bary, prim) as kk_closest_hit: a triangle counts exactly
 * when the ray passes the reference's slab test of the leaf's parent box (which, boxes being exact
 * unions, implies the tests of all nodes above it; for rays with a zero direction component every
 * box up to the root is tested) and the triangle arithmetic is the same.
 * Rays are traced in Morton order of their origins unless a flag says otherwise; outputs are indexed
 * like the input. The accel keeps a pointer to d_lbvh's internal_nodes: rebuild it when the
 * tree's arrays move. */
enum {
    KK_TRACE_UNSORTED = 1u,   /* trace in input order */
    KK_TRACE_REUSE_ORDER = 2u, /* the same rays as the previous call with this workspace: keep its order */
    KK_TRACE_WATERTIGHT = 4u,  /* watertight ray-triangle test (Woop, Benthin, Wald 2013) instead of the
                                  reference's Moeller-Trumbore: hits may differ on grazing rays */
    KK_TRACE_ENTRY_BOUND = 8u  /* a group of pending subtree slots carries a 16-bit lower bound of their
                                  entry distance and is dropped unvisited once that lies behind the best
                                  hit: same hits, fewer record visits, more instructions per visit (not
                                  with KK_TRACE_WATERTIGHT) */
};
size_t kk_trace_accel_bytes(uint32_t face_num, uint32_t compacted_face_num);
/* Where the pieces sit inside the accel buffer (byte offsets), for tools and tests that inspect it.
 * header: 6 floats root box, uint32 face_num, uint32 binary node count, the nodes pointer, uint32
 * wide node count (valid after the build). nodes: 128-byte records: float origin[3]; int8 exp[3];
 * uint8 imask; uint32 childBase, triBase, leafMask (bit 8: a leaf has more than 15 triangles, use
 * off16 / cnt16; bit 9: they do not fit 16 bits, counts holds the triangles of all leaf slots),
 * counts (4 bits per slot); uint8 q[6][8] (minx maxx miny maxy minz maxz per slot); uint16 off16[8],
 * cnt16[8]; 16 spare bytes.
 * tris: 48-byte records: float v0[3]; uint32 face; float v1[3]; uint32 parent (binary node);
 * float v2[3]; uint32 k (sorted position). parents: uint32 per binary node. */
typedef struct kk_trace_accel_layout_info {
    uint64_t header, nodes, tris, parents, total;
} kk_trace_accel_layout_info;
int kk_trace_accel_layout(uint32_t face_num, uint32_t compacted_face_num, kk_trace_accel_layout_info *out);
/* d_accel: 256-byte aligned device buffer of kk_trace_accel_bytes(). */
int kk_trace_accel_build(const kk_lbvh_device_data *d_lbvh, const float *d_positions,
                         const uint32_t *d_face_pos_idx, uint32_t face_num,
                         uint32_t compacted_face_num, void *d_accel, size_t accel_bytes, void *stream);
/* The same with the per-leaf boxes the LBVH build left in its workspace (d_leaf_aabbs float[M][6] at
 * kk_lbvh_workspace_layout().leaf_aabbs, valid until that workspace is reused): saves the pass that
 * gathers the vertices of every leaf again. d_leaf_aabbs == NULL is kk_trace_accel_build. */
int kk_trace_accel_build_with_leaf_boxes(const kk_lbvh_device_data *d_lbvh, const float *d_positions,
                                         const uint32_t *d_face_pos_idx, uint32_t face_num,
                                         uint32_t compacted_face_num, const float *d_leaf_aabbs,
                                         void *d_accel, size_t accel_bytes, void *stream);
/* Waits for `stream` and reads the build's status back: KK_ERR_DEVICE when the tree has a path of more
 * than 32 wide levels (the walks' stacks are made for that many; a tree kk_lbvh_build made never has,
 * its depth being bounded by the 30 Morton bits) or the buffer holds no tree of these sizes. */
int kk_trace_accel_check(const void *d_accel, uint32_t face_num, uint32_t compacted_face_num, void *stream);
size_t kk_closest_hit_accel_workspace_bytes(uint32_t ray_num);
/* d_counters (optional, uint64[16], zeroed by the caller): [0] 128-byte node records visited,
 * [1] triangles tested, [2] rays traced; diagnostics of the kernel's lane use: [3] passes of the wave
 * loop, [4] lanes that visited a record in them, [5] triangle phases, [6] lanes in them, [7] visits
 * that found no slot, [8] of those the ones that only the distance limit emptied, [9] used slots of
 * the visited records, [10] visits of records that hold leaves only, [11] slots dropped unvisited
 * (KK_TRACE_ENTRY_BOUND); wall clock in 10 ns ticks: [12] 2^62 - the earliest start of a wave, [13] 2^62 -
 * the earliest moment a wave found every ray queue used up, [14] the latest exit of a wave, [15] the sum of
 * the waves' lives. */
int kk_closest_hit_accel(const void *d_accel, uint32_t face_num, uint32_t compacted_face_num,
                         const float *d_rays, uint32_t ray_num, float *d_t, float *d_bary,
                         uint32_t *d_prim, uint32_t flags, unsigned long long *d_counters,
                         void *d_workspace, size_t workspace_bytes, void *stream);
/* Waits for `stream` and reports how the last kk_closest_hit_accel with this workspace ordered its
 * rays: KK_ERR_DEVICE when the ray sort gave up (kk_sort_workspace_status). The kernel sees that too
 * and then traces the rays in input order, so the hits are right either way; the status tells a caller
 * that the call ran at the unsorted rate. */
int kk_closest_hit_accel_status(const void *d_workspace, uint32_t ray_num, void *stream);

/* ================================================================================================
 * Path 1c: path tracer  --  kouek::RayTracer::RayTracer::Render(surface, size, RenderTarget::Scene)
 *          (src/ray_tracer/ray_tracer_render.cu:433-647; setters in src/ray_tracer/ray_tracer.cu)
 * ============================================================================================== */

/* include/ray_tracer/ray_tracer.h:29-33 (16-byte aligned: 32 bytes) */
typedef struct kk_pt_triangle {
    uint32_t norm_idx[3];
    uint32_t tex_coord_idx[3];
    uint32_t grp_idx;
    uint32_t _pad;
} kk_pt_triangle;

/* include/ray_tracer/material.h:12-22, same offsets (64 bytes) */
typedef struct kk_material {
    uint8_t emissive;
    uint8_t _pad[3];
    float emission[3], kd[3], ks[3], tr[3];
    float ni, ns, kd_over_kd_add_ks;
} kk_material;

/* include/ray_tracer/material.h:51-66, same offsets (80 bytes). type: 0 = quad, 1 = sphere */
typedef struct kk_light {
    uint8_t type;
    uint8_t _pad[3];
    union {
        struct {
            float o[3], u[3], v[3], n[3];
            float inv_area;
        } quad;
        struct {
            float o[3];
            float r, r_sqr;
        } sphere;
    };
    float area;
    float radiance[3];
    float _pad2[2];
} kk_light;

/* The fields of RayTracer::RenderParameter (ray_tracer.h:34-49) the Scene target reads; a host struct
 * passed by value to the kernel. Matrices are column-major (glm). Pointers are device pointers. */
typedef struct kk_pt_scene {
    uint8_t display_tree_height;
    uint8_t max_path_depth;
    uint8_t _pad[2];
    uint32_t light_num;
    float w2s[16];
    float inv_proj[16];
    const kk_pt_triangle *triangles;
    const uint32_t *tri_to_position_indices; /* uint32[N][3] */
    const float *positions;                  /* float[V][3] */
    const float *normals;                    /* float[.][3] */
    const float *tex_coords;                 /* float[.][2], unused by the Scene target */
    const kk_light *lights;
    const kk_material *materials;
    float light_tot_area;
    uint32_t _pad2;
    /* Optional: the trace-side tree of the same LBVH (kk_trace_accel_build) and the sizes it was
     * built with. When set, closest hits and shadow rays of kk_pt_render_sample walk it instead of
     * the binary tree; the image is the same. */
    const void *trace_accel;
    uint32_t accel_face_num, accel_compacted_face_num;
} kk_pt_scene;

/* RayTracer::RenderParameterPerFrame (ray_tracer.h:50-53); eye_rot column-major mat3 */
typedef struct kk_pt_frame {
    float eye_pos[3];
    float eye_rot[9];
} kk_pt_frame;

/* One Render(Scene) call restricted to pixel rows [y0, y1) (row 0 = bottom row, as in the reference's
 * GL surface): adds sample number `sample_index` (the reference's currAccuSamplePerPixel, 1-based) to
 * the running mean d_accum float[H][W][3] and writes d_rgba uint8[H][W][4] (alpha 255). The caller
 * zeroes d_accum when accumulation restarts (ray_tracer_render.cu:561-574). A pixel's random stream
 * depends only on (seed, y*W+x, sample_index), so any split of the rows gives the same image. */
int kk_pt_render_sample(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                        const kk_pt_frame *h_frame, float *d_accum, uint8_t *d_rgba, int width,
                        int height, int y0, int y1, float sample_index, uint32_t seed, void *stream);

/* sample_count consecutive calls of kk_pt_render_sample (sample_index, sample_index + 1, ...) as one
 * launch: a lane renders the samples of its pixel one after the other, so the running mean of every
 * pixel is folded in the same order and the result equals the separate calls bit for bit, while the
 * GPU drains once per call instead of once per sample. The reference renders one sample per Render()
 * (ray_tracer_render.cu:561-638); this is for callers that know they want several. */
int kk_pt_render_samples(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                         const kk_pt_frame *h_frame, float *d_accum, uint8_t *d_rgba, int width,
                         int height, int y0, int y1, float sample_index, int sample_count, uint32_t seed,
                         void *stream);

/* The same samples for the 16-row stripes stripe_first, stripe_first + stripe_stride, ... of the image
 * (stripe s = rows [16 s, 16 s + 16)): the share of one GPU when the frame is dealt out to several in
 * round-robin stripes. The j-th stripe of the launch goes to rows [16 j, 16 j + 16) of the compact arrays
 * d_accum_stripes float[kk_stripes_per_rank][16][W][3] and d_stripes uint8[...][16][W][4] - the latter is
 * the block a rank hands to kk_gather_stripes; rows of a last, partial stripe beyond the image are not
 * written. A pixel's random stream is keyed by its place in the image, so the block holds the bytes the
 * same rows have after kk_pt_render_samples on one GPU. No counterpart in the reference, which renders on
 * one GPU (ray_tracer_render.cu:579-638 is the work that is dealt out). */
int kk_pt_render_stripes_compact(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                                 const kk_pt_frame *h_frame, float *d_accum_stripes, uint8_t *d_stripes,
                                 int width, int height, int stripe_first, int stripe_stride,
                                 float sample_index, int sample_count, uint32_t seed, void *stream);

/* Diagnostics of the path-tracing kernel on the trace-side tree: while d_counters (uint64[16], zeroed by
 * the caller) is set, every kk_pt_render_sample adds [0] passes of the waves' walk loop, [1] lanes that
 * walked in them, [2] runs of a shading section, [3] lanes they ran for, [4] rays (closest-hit and
 * shadow), [5] node records visited, [6] triangles tested, [7] rounds of the pooled triangle test,
 * [15] pairs tested in them, [14] triangles of the hit leaf slots alone, [8..13] clock cycles of one lane
 * per wave in: walk loop control, record visits, triangle rounds, light samples, next directions, pixel
 * write + refill. NULL switches it off (the default). */
int kk_pt_set_work_counters(unsigned long long *d_counters);

/* The other targets of RayTracer::Render (ray_tracer.h RenderTarget; ray_tracer_render.cu:461-558):
 * one frame of rows [y0, y1) straight into d_rgba, no accumulation. AABBs draws the boxes at tree depth
 * h_scene->display_tree_height (:8-100); Triangles / Positions / Normals / TextureCoords shade the
 * closest hit (:508-551, including the reference's use of norms[0] for two weights and of the normals
 * array under TextureCoords); Lights draws the light shapes (:272-310). */
enum {
    KK_RT_TARGET_AABBS = 1,
    KK_RT_TARGET_TRIANGLES = 2,
    KK_RT_TARGET_LIGHTS = 3,
    KK_RT_TARGET_POSITIONS = 4,
    KK_RT_TARGET_NORMALS = 5,
    KK_RT_TARGET_TEXTURE_COORDS = 6
};
int kk_rt_render_target(const kk_lbvh_device_data *d_lbvh, const kk_pt_scene *h_scene,
                        const kk_pt_frame *h_frame, int target, uint8_t *d_rgba, int width, int height,
                        int y0, int y1, void *stream);

/* The first `count` uniforms in (0, 1] of the stream (seed, pixel, sample), for checking the generator. */
int kk_rng_uniforms(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t count, float *d_out,
                    void *stream);

/* ================================================================================================
 * Path 2a: Depth-Box VDB build  --  kouek::RayCaster::DepthBoxVDB::BuildFrom<T>
 *          (src/ray_caster/dpbx_vdb.cu:275-610, updateAtlas :117-273), T = uint8_t or uint16_t
 * ============================================================================================== */

#define KK_VDB_MAX_LEVEL_NUM 3
#define KK_VDB_UNDEF_RLT_IDX 0xFFFFFFFFu

/* include/ray_caster/dpbx_vdb.h:39-54, same offsets (128 bytes) */
typedef struct kk_vdb_param {
    uint8_t root_lev;
    uint8_t apron_wid;
    uint8_t apron_dep_wid;
    uint8_t log2_dims[KK_VDB_MAX_LEVEL_NUM];
    int32_t dims[KK_VDB_MAX_LEVEL_NUM];
    int32_t vox_per_atlas_brick;
    int32_t min_dep_pos_val_in_brick;
    int32_t max_dep_pos_val_in_brick;
    int32_t atlas_brick_per_vdb[3];
    uint64_t node_starts[KK_VDB_MAX_LEVEL_NUM];
    uint64_t child_starts[KK_VDB_MAX_LEVEL_NUM];
    float voxs_per_child[KK_VDB_MAX_LEVEL_NUM];
    float vox_per_vdb;
    float vox_per_vol[3];
    uint32_t _pad;
} kk_vdb_param;

/* include/ray_caster/dpbx_vdb.h:55-59 (32 bytes) */
typedef struct kk_vdb_node {
    int32_t pos[3];                /* first voxel of the node in the volume */
    int32_t brick_pos_in_atlas[3]; /* leaves only */
    uint64_t child_pool_offs_relative;
} kk_vdb_node;

/* Counterpart of DepthBoxVDB::DeviceData (dpbx_vdb.h:60-67), same size (176 bytes). The three CUDA
 * surface / texture handles are replaced by the atlas in linear memory, stored brick by brick: brick k
 * belongs to leaf node k and holds vox_per_atlas_brick^3 voxels, x fastest. */
typedef struct kk_vdb_device_data {
    kk_vdb_node *node_pools;
    uint32_t *child_pools;
    uint32_t *atlas_brick_to_node_indices;
    void *atlas;
    uint64_t leaf_num;
    uint64_t voxel_bytes; /* 1 or 2 */
    kk_vdb_param vdb_param;
} kk_vdb_device_data;

/* Sizes the host wiring step (kk_vdb_plan) reports. */
typedef struct kk_vdb_plan {
    kk_vdb_param param;
    uint32_t leaf_slots[3]; /* bricks per axis covering the volume */
    uint32_t leaf_num;      /* valid (non-empty) leaves */
    uint64_t node_num;      /* entries of node_pools (root last) */
    uint64_t child_num;     /* entries of child_pools */
    uint64_t atlas_map_num; /* entries of atlas_brick_to_node_indices */
    uint64_t atlas_voxels;  /* leaf_num * vox_per_atlas_brick^3 */
} kk_vdb_plan;

/* Step 1 (device): d_leaf_valid[z*ly*lx + y*lx + x] = 0 when every voxel of the brick (voxels outside
 * the volume read 0) lies in an empty scalar range, else 1 (dpbx_vdb.cu:345-385; like the reference it
 * counts (voxel, range) pairs). d_volume is T[dz][dy][dx], x fastest. h_empty_rngs: n_rngs pairs of T. */
int kk_vdb_classify_u8(const uint8_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                       uint32_t log2_dim0, const uint8_t *h_empty_rngs, uint32_t n_rngs,
                       uint32_t *d_leaf_valid, void *stream);
int kk_vdb_classify_u16(const uint16_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                        uint32_t log2_dim0, const uint16_t *h_empty_rngs, uint32_t n_rngs,
                        uint32_t *d_leaf_valid, void *stream);

/* Step 2 (host only, no GPU call): parameters and pool sizes from the validity flags
 * (dpbx_vdb.cu:282-338, 387-467). Fails like the reference when a log2 dimension is outside [1, 10] or
 * three levels cannot cover the volume, and also when one brick covers the whole volume or no leaf is
 * valid (the reference then builds a tree its own marcher cannot walk). */
int kk_vdb_plan_build(uint32_t dx, uint32_t dy, uint32_t dz, const uint8_t log2_dims[3],
                      uint8_t apron_wid, int use_depth_box, const uint32_t *h_leaf_valid,
                      kk_vdb_plan *out);
/* Optional, between steps 2 and 4 (device): the leaf nodes' `pos` fields made on the GPU from the same
 * validity flags (leaf k = the k-th valid brick in the pools' (x, y, z) order), into an array of
 * plan->leaf_num nodes of the caller's. kk_vdb_fill_atlas_* reads nothing else of the nodes it is given,
 * so with this array in place of d_node_pools the atlas can be filled while the host still runs step 3.
 * No counterpart in the reference, which wires on the host first (dpbx_vdb.cu:173-330). */
int kk_vdb_leaf_nodes(const uint32_t *d_leaf_valid, const kk_vdb_plan *plan, kk_vdb_node *d_leaf_nodes,
                      void *stream);
/* Step 3 (host only): fills node pools (sorted by level, then x, y, z; root last), dense child pools
 * (index (z << 2L) | (y << L) | x, undefined = 0xFFFFFFFF) and the atlas map (dpbx_vdb.cu:57-115,
 * 404-445, 524-571) into caller-allocated host arrays of the planned sizes. */
int kk_vdb_wire(const kk_vdb_plan *plan, const uint32_t *h_leaf_valid, kk_vdb_node *h_node_pools,
                uint32_t *h_child_pools, uint32_t *h_atlas_brick_to_node);

/* Step 4 (device): fills the brick-major atlas for the plan's leaves (dpbx_vdb.cu:176-272): volume
 * voxels for the brick and its apron, depth values on the one-face shell voxels, 0 on shell edges. */
int kk_vdb_fill_atlas_u8(const uint8_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                         const kk_vdb_plan *plan, const kk_vdb_node *d_node_pools,
                         const uint8_t *h_empty_rngs, uint32_t n_rngs, uint8_t *d_atlas, void *stream);
int kk_vdb_fill_atlas_u16(const uint16_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz,
                          const kk_vdb_plan *plan, const kk_vdb_node *d_node_pools,
                          const uint16_t *h_empty_rngs, uint32_t n_rngs, uint16_t *d_atlas,
                          void *stream);

/* ================================================================================================
 * Path 2b: DVR marcher  --  kouek::RayCaster::RayCaster::RenderDepthBoxVDB<T>
 *          (src/ray_caster/ray_caster_dpbx_vdb.cu:439-532)
 * ============================================================================================== */

/* The fields of RayCaster::RenderParameter (include/ray_caster/ray_caster.h:18-28) the marcher reads */
typedef struct kk_dvr_params {
    float w2s[16];      /* column-major */
    float inv_proj[16]; /* column-major */
    float max_alpha;
    float sampling_rate;
    int32_t use_shading;
    float ka, kd, ks, shiness;
    float light_pos[3]; /* in volume (voxel) space */
    float light_col[3];
} kk_dvr_params;

/* RayCaster::RenderParameterPerFrame (ray_caster.h:29-32) */
typedef struct kk_dvr_frame {
    float eye_pos[3];
    float eye_rot[9];
} kk_dvr_frame;

/* One frame restricted to pixel rows [y0, y1) (row 0 = bottom row): d_rgba uint8[H][W][4] with alpha =
 * accumulated opacity. h_vdb is the HOST copy of the device-data struct (its pointers are device
 * pointers). d_tf float[256][4]. d_counters (optional, uint64[2], zeroed by the caller) accumulates
 * samples taken and leaf bricks entered. Fails (KK_ERR_INVALID) when a voxs_per_child entry is not a
 * power of two: kk_vdb_plan makes them products of the 2^log2_dims, and the marcher scales by them
 * where the reference divides, which is the same value for powers of two only. */
int kk_dvr_render_u8(const kk_vdb_device_data *h_vdb, const float *d_tf, const kk_dvr_params *h_params,
                     const kk_dvr_frame *h_frame, uint8_t *d_rgba, int width, int height, int y0,
                     int y1, unsigned long long *d_counters, void *stream);
int kk_dvr_render_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                      const kk_dvr_params *h_params, const kk_dvr_frame *h_frame, uint8_t *d_rgba,
                      int width, int height, int y0, int y1, unsigned long long *d_counters,
                      void *stream);
/* The same frame with its 16 x 16 pixel tiles started in a given order, and what each tile took reported
 * back: the frame time of a marcher is set by where its longest rays are started, and an interactive
 * renderer knows that from the frame before. Tiles are numbered row by row over the rendered rows
 * (kk_dvr_tile_count(width, y1 - y0) of them). d_tile_order (optional): a permutation of the tile numbers,
 * workgroup n renders tile d_tile_order[n]. d_tile_cost (optional): uint32 per tile, the time the tile's
 * workgroup took, in units of 64 clock cycles, capped at 65535. kk_dvr_tile_order sorts the tiles by cost
 * class (256 classes, sixteen to a doubling of the cost), the costly ones first, ties by tile number, into
 * d_tile_order. The image does not depend on the order.
 * No counterpart in the reference, whose launch order is the tiles' own. The buffers belong to one
 * sequence of frames: frames in flight on different streams need a set each. */
uint32_t kk_dvr_tile_count(int width, int rows);
int kk_dvr_render_ordered_u8(const kk_vdb_device_data *h_vdb, const float *d_tf, const kk_dvr_params *h_params,
                             const kk_dvr_frame *h_frame, uint8_t *d_rgba, int width, int height, int y0,
                             int y1, uint32_t *d_tile_cost, const uint32_t *d_tile_order, void *stream);
int kk_dvr_render_ordered_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                              const kk_dvr_params *h_params, const kk_dvr_frame *h_frame, uint8_t *d_rgba,
                              int width, int height, int y0, int y1, uint32_t *d_tile_cost,
                              const uint32_t *d_tile_order, void *stream);
/* The same for the compact stripes of one GPU (kk_dvr_render_stripes_compact_* below): the tiles are those
 * of this GPU's stripes, row by row, kk_dvr_stripe_tile_count of them. */
uint32_t kk_dvr_stripe_tile_count(int width, int height, int stripe_first, int stripe_stride);
int kk_dvr_render_stripes_compact_ordered_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                             const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                             uint8_t *d_stripes, int width, int height, int stripe_first,
                                             int stripe_stride, uint32_t *d_tile_cost,
                                             const uint32_t *d_tile_order, void *stream);
int kk_dvr_render_stripes_compact_ordered_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                              const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                              uint8_t *d_stripes, int width, int height, int stripe_first,
                                              int stripe_stride, uint32_t *d_tile_cost,
                                              const uint32_t *d_tile_order, void *stream);
size_t kk_dvr_tile_order_workspace_bytes(uint32_t tiles);
int kk_dvr_tile_order(const uint32_t *d_tile_cost, uint32_t tiles, uint32_t *d_tile_order, void *d_workspace,
                      size_t workspace_bytes, void *stream);
/* Waits for `stream` and reports on the last kk_dvr_tile_order with this workspace: KK_ERR_DEVICE when
 * its sort gave up (kk_sort_workspace_status). d_tile_order then holds the launch order 0, 1, 2, ...
 * (written on the stream after the sort), never a non-permutation, so no tile is left unrendered.
 * (Only frames of more than 2^18 tiles go through that sort; the order of smaller ones is made by one
 * launch that waits for nothing, and the status is then always KK_OK.) */
int kk_dvr_tile_order_status(const void *d_workspace, uint32_t tiles, void *stream);
/* The same frame restricted to the 16-row stripes stripe_first, stripe_first + stripe_stride, ... of
 * the image (stripe s = rows [16 s, 16 s + 16)): the share of one GPU when the image is dealt out to
 * several in round-robin stripes. No counterpart in the reference, which renders on one GPU. */
int kk_dvr_render_stripes_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                             const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                             uint8_t *d_rgba, int width, int height, int stripe_first,
                             int stripe_stride, unsigned long long *d_counters, void *stream);
int kk_dvr_render_stripes_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                              const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                              uint8_t *d_rgba, int width, int height, int stripe_first,
                              int stripe_stride, unsigned long long *d_counters, void *stream);

/* The same stripes written one after the other into d_stripes uint8[kk_stripes_per_rank][16][W][4]
 * (the j-th stripe of this launch, image stripe stripe_first + j * stripe_stride, goes to rows
 * [16 j, 16 j + 16) of d_stripes): the contiguous block a rank hands to kk_gather_stripes. Rows of a
 * last, partial stripe beyond the image are not written. */
int kk_dvr_render_stripes_compact_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                     const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                     uint8_t *d_stripes, int width, int height, int stripe_first,
                                     int stripe_stride, unsigned long long *d_counters, void *stream);
int kk_dvr_render_stripes_compact_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                                      const kk_dvr_params *h_params, const kk_dvr_frame *h_frame,
                                      uint8_t *d_stripes, int width, int height, int stripe_first,
                                      int stripe_stride, unsigned long long *d_counters, void *stream);

/* RenderDepthBoxVDB with any DPBXVDBRenderTarget (include/ray_caster/ray_caster.h:85; kernels
 * src/ray_caster/ray_caster_dpbx_vdb.cu:284-437): AABBs colours the nodes of level display_lev
 * (RayCaster::SetDepthBoxVDBDisplayLevel, 0 = leaves ... root_lev), Depths and
 * PositionsAfterDepthSkip inspect the display_leaf_layer-th leaf along each ray
 * (SetDepthBoxVDBDisplayLeafLayer). Scene forwards to kk_dvr_render_*; d_tf is only read for it. */
enum {
    KK_VDB_TARGET_SCENE = 0,
    KK_VDB_TARGET_AABBS = 1,
    KK_VDB_TARGET_DEPTHS = 2,
    KK_VDB_TARGET_POSITIONS_AFTER_DEPTH_SKIP = 3
};
int kk_dvr_render_target_u8(const kk_vdb_device_data *h_vdb, const float *d_tf,
                            const kk_dvr_params *h_params, const kk_dvr_frame *h_frame, int target,
                            int display_lev, int display_leaf_layer, uint8_t *d_rgba, int width,
                            int height, int y0, int y1, void *stream);
int kk_dvr_render_target_u16(const kk_vdb_device_data *h_vdb, const float *d_tf,
                             const kk_dvr_params *h_params, const kk_dvr_frame *h_frame, int target,
                             int display_lev, int display_leaf_layer, uint8_t *d_rgba, int width,
                             int height, int y0, int y1, void *stream);

/* ================================================================================================
 * Path 2c: RAW-volume ray caster  --  kouek::RayCaster::RayCaster::RenderRAWVolume
 *          (src/ray_caster/ray_caster_raw_volume.cu:27-246): the same compositing over the dense volume,
 *          no empty-space skipping. d_volume is T[dz][dy][dx]; samples are fp32 trilinear reads with
 *          0 outside the volume. d_counters (optional, uint64[1]) accumulates the samples taken.
 * ============================================================================================== */
enum { KK_RAW_TARGET_SCENE = 0, KK_RAW_TARGET_AABB = 1 }; /* RayCaster::RAWVolumeRenderTarget */
int kk_raw_render_u8(const uint8_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz, const float *d_tf,
                     const kk_dvr_params *h_params, const kk_dvr_frame *h_frame, int target,
                     uint8_t *d_rgba, int width, int height, int y0, int y1,
                     unsigned long long *d_counters, void *stream);
int kk_raw_render_u16(const uint16_t *d_volume, uint32_t dx, uint32_t dy, uint32_t dz, const float *d_tf,
                      const kk_dvr_params *h_params, const kk_dvr_frame *h_frame, int target,
                      uint8_t *d_rgba, int width, int height, int y0, int y1,
                      unsigned long long *d_counters, void *stream);

/* ================================================================================================
 * Several GPUs (one process per GPU, RCCL over xGMI). No counterpart in the reference, which renders on
 * one GPU: the structure (LBVH + mesh, or VDB + atlas) is built once on one rank and broadcast, every
 * rank renders its share of the image (round-robin 16-row stripes) or traces its share of the rays,
 * and the stripes are gathered to one rank. RCCL is opened at run time by the first kk_comm_* call.
 * ============================================================================================== */
typedef struct kk_comm kk_comm;
#define KK_COMM_ID_BYTES 128
/* Rank 0 makes the id and hands it to the other ranks by any means the host has (a file, MPI, a
 * socket); every rank then calls kk_comm_init with its own GPU current (kk_set_device). */
int kk_comm_unique_id(void *h_id /* KK_COMM_ID_BYTES */);
int kk_comm_init(int nranks, int rank, const void *h_id, kk_comm **out);
/* Waits for everything queued on the communicator's device (collectives may still be in flight on
 * side streams) before it frees the communicator. The kk_comm_* calls that issue work fail with
 * KK_ERR_INVALID when the current device is not the one the communicator was made on. */
int kk_comm_destroy(kk_comm *comm);
int kk_comm_rank(const kk_comm *comm, int *rank, int *nranks);
/* One collective over `bytes` of device memory, enqueued on `stream`. */
int kk_comm_broadcast(kk_comm *comm, void *d_buf, size_t bytes, int root, void *stream);
/* A few host bytes (sizes, a plan; at most 4096) from root to every rank; waits for `stream`. */
int kk_comm_broadcast_host(kk_comm *comm, void *h_buf, size_t bytes, int root, void *stream);
/* Every rank's bytes_per_rank bytes land at d_recv + rank * bytes_per_rank on root (d_recv may be NULL
 * elsewhere): one grouped receive per peer, i.e. one direct link each. Enqueued on `stream`. */
int kk_comm_gather(kk_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, int root,
                   void *stream);

/* An LBVH and the mesh it indexes in ONE device allocation, so that it travels in one collective:
 * the 64-byte kk_lbvh_device_data first, then positions float[V][3], face_pos_idx uint32[N][3], faces
 * uint32[N], compacted uint32[N], nodes kk_lbvh_node[N], each 256-byte aligned. Point kk_lbvh_build at
 * these pieces (d_out = blob + device_data). */
typedef struct kk_lbvh_blob_layout {
    uint64_t device_data, positions, face_pos_idx, faces, compacted, nodes, total;
} kk_lbvh_blob_layout;
int kk_lbvh_blob_layout_of(uint32_t face_num, uint32_t vert_num, kk_lbvh_blob_layout *out);
/* Rewrites the three pointers of the blob's kk_lbvh_device_data for this blob's own address (device
 * pointers do not travel); the counts and the root box stay. Enqueued on `stream`. */
int kk_lbvh_blob_bind(void *d_blob, uint32_t face_num, uint32_t vert_num, void *stream);
/* kk_comm_broadcast of the whole blob, then kk_lbvh_blob_bind on every rank. face_num and vert_num
 * must be known on every rank (kk_comm_broadcast_host). */
int kk_lbvh_broadcast(kk_comm *comm, void *d_blob, uint32_t face_num, uint32_t vert_num, int root,
                      void *stream);

/* The arrays RayTracer::SetMesh uploads (src/ray_tracer/ray_tracer.cu:15-82) in ONE allocation, so that
 * the scene travels beside the tree in one collective: triangles kk_pt_triangle[face_num],
 * tri_to_position_indices uint32[face_num][3], positions float[vert_num][3], normals float[normal_num][3],
 * tex_coords float[tex_coord_num][2], lights kk_light[light_num], materials kk_material[material_num],
 * each 256-byte aligned. The sizes (32 bytes) must be known on every rank (kk_comm_broadcast_host). */
typedef struct kk_pt_scene_sizes {
    uint32_t face_num, vert_num, normal_num, tex_coord_num, light_num, material_num;
    float light_tot_area;
    uint32_t _pad;
} kk_pt_scene_sizes;
typedef struct kk_pt_scene_blob_layout {
    uint64_t triangles, tri_to_position_indices, positions, normals, tex_coords, lights, materials, total;
} kk_pt_scene_blob_layout;
int kk_pt_scene_blob_layout_of(const kk_pt_scene_sizes *sizes, kk_pt_scene_blob_layout *out);
/* Host-side: points h_io's seven array pointers into this blob and sets light_num / light_tot_area; the
 * other fields of h_io (matrices, depth, trace tree) are left alone. */
int kk_pt_scene_blob_bind(void *d_blob, const kk_pt_scene_sizes *sizes, kk_pt_scene *h_io);
/* One collective over the blob. */
int kk_pt_scene_broadcast(kk_comm *comm, void *d_blob, const kk_pt_scene_sizes *sizes, int root, void *stream);

/* A Depth-Box VDB in one allocation: node pools, child pools, atlas map, atlas (256-byte aligned
 * pieces, sizes from the plan). */
typedef struct kk_vdb_blob_layout {
    uint64_t node_pools, child_pools, atlas_map, atlas, total;
} kk_vdb_blob_layout;
int kk_vdb_blob_layout_of(const kk_vdb_plan *plan, uint32_t voxel_bytes, kk_vdb_blob_layout *out);
/* Host-side: the device-data struct (pointers into this blob, parameters from the plan). */
int kk_vdb_blob_device_data(void *d_blob, const kk_vdb_plan *plan, uint32_t voxel_bytes,
                            kk_vdb_device_data *h_out);
/* One collective over the blob. The plan (176 bytes) must be known on every rank first. */
int kk_vdb_broadcast(kk_comm *comm, void *d_blob, const kk_vdb_plan *plan, uint32_t voxel_bytes, int root,
                     void *stream);

/* Image stripes: stripe s = rows [16 s, 16 s + 16); rank r renders stripes r, r + nranks, ... into a
 * compact block of kk_stripes_per_rank(height, nranks) stripes (kk_dvr_render_stripes_compact_*).
 * kk_gather_stripes collects the blocks on root (d_blocks uint8[nranks][per][16][W][4], root only);
 * kk_unstripe_rgba puts them back in image order: d_rgba uint8[H][W][4]. */
int kk_stripes_per_rank(int height, int nranks);
int kk_gather_stripes(kk_comm *comm, const uint8_t *d_stripes, uint8_t *d_blocks, int width, int height,
                      int root, void *stream);
int kk_unstripe_rgba(const uint8_t *d_blocks, int nranks, int width, int height, uint8_t *d_rgba,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KOUEK_HIP_H */
