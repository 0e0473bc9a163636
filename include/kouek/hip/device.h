// RAII device memory and the headless replacements of the reference's CUDA-side resources
// (include/cuda/texture.h: Array / Texture / Surface), all on top of the C ABI.
#ifndef KOUEK_HIP_DEVICE_H
#define KOUEK_HIP_DEVICE_H

#include <array>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <vector>

#include "../../kouek_hip.h"
#include "../math_types.h"

namespace kouek {

struct Noncopyable {
    Noncopyable() = default;
    Noncopyable(const Noncopyable &) = delete;
    Noncopyable &operator=(const Noncopyable &) = delete;
};

// include/util/meta_type.h:12-28
template <typename T> struct Modifiable {
    bool modified = true;
    T dat{};
    template <typename U> void Set(U T::*member, const U &val) {
        if (std::memcmp(&(dat.*member), &val, sizeof(U)) != 0) {
            dat.*member = val;
            modified = true;
        }
    }
    const T &GetAndReset() {
        modified = false;
        return dat;
    }
};

namespace HIP {

// The reference's KOUEK_CUDA_CHECK prints and carries on (include/cuda/helper.h:14-26).
inline bool Check(int rc, const char *tag, const char *file, int line) {
    if (rc == KK_OK)
        return true;
    std::cerr << tag << " at " << file << ":" << line << ". " << kk_last_error() << "\n";
    return false;
}
#define KOUEK_HIP_CHECK(tag, call) kouek::HIP::Check((call), tag, __FILE__, __LINE__)

template <typename T> class DeviceBuffer : Noncopyable {
    T *ptr = nullptr;
    size_t count = 0;

  public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t n) { resize(n); }
    explicit DeviceBuffer(const std::vector<T> &host) { assign(host); }
    ~DeviceBuffer() { clear(); }
    DeviceBuffer(DeviceBuffer &&o) noexcept : ptr(o.ptr), count(o.count) { o.ptr = nullptr, o.count = 0; }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
        if (this != &o) {
            clear();
            ptr = o.ptr, count = o.count;
            o.ptr = nullptr, o.count = 0;
        }
        return *this;
    }
    void clear() {
        if (ptr)
            kk_free(ptr);
        ptr = nullptr;
        count = 0;
    }
    bool resize(size_t n) {
        clear();
        if (n == 0)
            return true;
        void *p = nullptr;
        if (kk_malloc(&p, n * sizeof(T)) != KK_OK)
            return false;
        ptr = static_cast<T *>(p);
        count = n;
        return true;
    }
    bool assign(const std::vector<T> &host) {
        if (!resize(host.size()))
            return false;
        return host.empty() || kk_memcpy_h2d(ptr, host.data(), host.size() * sizeof(T), nullptr) == KK_OK;
    }
    bool zero() { return count == 0 || kk_memset(ptr, 0, count * sizeof(T), nullptr) == KK_OK; }
    std::vector<T> to_host(size_t n = SIZE_MAX) const {
        std::vector<T> h(n < count ? n : count);
        if (!h.empty())
            kk_memcpy_d2h(h.data(), ptr, h.size() * sizeof(T), nullptr);
        return h;
    }
    T *data() const { return ptr; }
    size_t size() const { return count; }
};

struct uchar4 {
    uint8_t x, y, z, w;
};

// A HIP stream of the application's own (not in the reference, whose calls all go to the default stream).
// The render calls take one optionally: frames rendered alternately on two streams into two frame buffers
// overlap, which hides the end of each frame, where only its longest rays are still marching.
class Stream : Noncopyable {
    void *s = nullptr;

  public:
    Stream() { kk_stream_create(&s); }
    ~Stream() { kk_stream_destroy(s); }
    void *Get() const { return s; }
    void Synchronize() const { kk_stream_synchronize(s); }
};

// Stands in for the cudaSurfaceObject_t of the reference's Render calls: RGBA8, row 0 = bottom row,
// pitch = 4 * width.
class FrameBuffer : Noncopyable {
    DeviceBuffer<uchar4> buf;
    ivec2 sz{0, 0};

  public:
    FrameBuffer() = default;
    explicit FrameBuffer(const ivec2 &size) { Resize(size); }
    void Resize(const ivec2 &size) {
        sz = size;
        buf.resize(static_cast<size_t>(size.x) * size.y);
        buf.zero();
    }
    const ivec2 &GetSize() const { return sz; }
    uint8_t *Get() const { return reinterpret_cast<uint8_t *>(buf.data()); }
    void ReadBack(std::vector<uchar4> &out) const { out = buf.to_host(); }
};

// Stands in for the point-filtered volume texture: voxels T[dz][dy][dx], x fastest.
template <typename T> class Volume : Noncopyable {
    DeviceBuffer<T> buf;
    uint32_t dim[3] = {0, 0, 0};

  public:
    Volume(const std::vector<T> &dat, uint32_t dx, uint32_t dy, uint32_t dz) : buf(dat) {
        dim[0] = dx, dim[1] = dy, dim[2] = dz;
    }
    bool IsComplete() const { return buf.size() == static_cast<size_t>(dim[0]) * dim[1] * dim[2] && buf.size() != 0; }
    const T *Get() const { return buf.data(); }
    const uint32_t *GetDimension() const { return dim; }
};

// One process per GPU: the library's RCCL communicator (kk_comm_*). Rank 0 calls MakeId() and hands
// the 128 bytes to the other ranks by whatever means the host program has; every rank then constructs
// its Communicator with its own GPU current (kk_set_device). Not part of the reference, which renders
// on one GPU.
class Communicator : Noncopyable {
    kk_comm *comm = nullptr;
    int rank = 0, nranks = 1;

  public:
    using Id = std::array<uint8_t, KK_COMM_ID_BYTES>;
    static bool MakeId(Id &id) { return KOUEK_HIP_CHECK("[Communicator Error]", kk_comm_unique_id(id.data())); }
    Communicator(int nranks, int rank, const Id &id) : rank(rank), nranks(nranks) {
        if (!KOUEK_HIP_CHECK("[Communicator Error]", kk_comm_init(nranks, rank, id.data(), &comm)))
            comm = nullptr;
    }
    ~Communicator() { kk_comm_destroy(comm); }
    bool IsComplete() const { return comm != nullptr; }
    kk_comm *Get() const { return comm; }
    int GetRank() const { return rank; }
    int GetRankNum() const { return nranks; }
};

// Stands in for the 1-D float4 transfer-function texture (256 entries).
class TransferFunctionTable : Noncopyable {
    DeviceBuffer<vec4> buf;

  public:
    explicit TransferFunctionTable(const std::vector<vec4> &table) : buf(table) {}
    bool IsComplete() const { return buf.size() == 256; }
    const float *Get() const { return reinterpret_cast<const float *>(buf.data()); }
};

} // namespace HIP
} // namespace kouek

#endif // KOUEK_HIP_DEVICE_H
