// fp32 vector helpers for the device code. The operation ORDER follows the glm 0.9.9.8 the reference
// computes with (3rd/glm/detail/func_common.inl:17-30, func_geometric.inl:12,53,75-77,88,
// type_mat4x4.inl:563-571, type_mat3x3.inl:468-474) and the library is built with
// -ffp-contract=off, so that Morton codes, boxes, hit distances and barycentrics carry the same bits as
// the reference's arithmetic. Division and sqrt are the correctly rounded ones (hipcc default).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KK_HD __host__ __device__ __forceinline__
#define KK_D __device__ __forceinline__

namespace kk {

constexpr float FloatMax = 3.402823466e+38f;
constexpr float FloatMin = -3.402823466e+38f;
constexpr float FloatEps = 1.192092896e-07f;

KK_HD float gmin(float x, float y) { return (y < x) ? y : x; }
KK_HD float gmax(float x, float y) { return (x < y) ? y : x; }

struct f2 {
    float x, y;
};
struct f3 {
    float x, y, z;
    KK_HD float &operator[](int i) { return (&x)[i]; }
    KK_HD const float &operator[](int i) const { return (&x)[i]; }
};
struct f4 {
    float x, y, z, w;
};
struct i3 {
    int x, y, z;
    KK_HD int &operator[](int i) { return (&x)[i]; }
    KK_HD const int &operator[](int i) const { return (&x)[i]; }
};
struct u3 {
    uint32_t x, y, z;
};

KK_HD f3 F3(float a) { return {a, a, a}; }
KK_HD f3 F3(float a, float b, float c) { return {a, b, c}; }
KK_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
KK_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
KK_HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
KK_HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
KK_HD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
KK_HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
KK_HD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
KK_HD f3 operator/(float s, f3 a) { return {s / a.x, s / a.y, s / a.z}; }
KK_HD f3 operator+(f3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
KK_HD f3 operator-(f3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
KK_HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }

KK_HD f4 operator*(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
KK_HD f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

KK_HD float dot(f3 a, f3 b) {
    f3 t = a * b;
    return t.x + t.y + t.z;
}
KK_HD f3 cross(f3 x, f3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
KK_HD float length(f3 v) { return sqrtf(dot(v, v)); }
KK_HD f3 normalize(f3 v) { return v * (1.f / sqrtf(dot(v, v))); }
KK_HD f3 vabs(f3 v) { return {fabsf(v.x), fabsf(v.y), fabsf(v.z)}; }
KK_HD f3 vfloor(f3 v) { return {floorf(v.x), floorf(v.y), floorf(v.z)}; }

// Column-major, like glm.
struct m3 {
    f3 c[3];
};
struct m4 {
    f4 c[4];
};
KK_HD f3 operator*(const m3 &m, f3 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
KK_HD f4 operator*(const m4 &m, f4 v) {
    f4 mul0 = m.c[0] * v.x;
    f4 mul1 = m.c[1] * v.y;
    f4 add0 = mul0 + mul1;
    f4 mul2 = m.c[2] * v.z;
    f4 mul3 = m.c[3] * v.w;
    f4 add1 = mul2 + mul3;
    return add0 + add1;
}

// include/cg/bounding_box.h:10-66
struct AABB {
    f3 minPos, maxPos;
    KK_HD static AABB Initial() { return {F3(FloatMax), F3(FloatMin)}; }
    KK_HD void Expand(f3 p) {
        if (minPos.x > p.x) minPos.x = p.x;
        if (maxPos.x < p.x) maxPos.x = p.x;
        if (minPos.y > p.y) minPos.y = p.y;
        if (maxPos.y < p.y) maxPos.y = p.y;
        if (minPos.z > p.z) minPos.z = p.z;
        if (maxPos.z < p.z) maxPos.z = p.z;
    }
    KK_HD void Expand(const AABB &o) {
        if (minPos.x > o.minPos.x) minPos.x = o.minPos.x;
        if (maxPos.x < o.maxPos.x) maxPos.x = o.maxPos.x;
        if (minPos.y > o.minPos.y) minPos.y = o.minPos.y;
        if (maxPos.y < o.maxPos.y) maxPos.y = o.maxPos.y;
        if (minPos.z > o.minPos.z) minPos.z = o.minPos.z;
        if (maxPos.z < o.maxPos.z) maxPos.z = o.maxPos.z;
    }
};

struct Ray {
    f3 pos, dir;
};

// A ray with a NaN or an infinite component: every comparison of the reference's slab test comes out
// "not missed" for it, so the reference walks the whole tree and finds nothing it can keep (a NaN
// distance is never the smaller one; only an infinite direction can leave it a spurious hit at t = 0).
// The walks here give such a ray a miss at once instead of visiting every node for it.
KK_HD bool ray_is_finite(const f3 &pos, const f3 &dir) {
    return fabsf(pos.x) <= FloatMax && fabsf(pos.y) <= FloatMax && fabsf(pos.z) <= FloatMax &&
           fabsf(dir.x) <= FloatMax && fabsf(dir.y) <= FloatMax && fabsf(dir.z) <= FloatMax;
}

// include/cg/ray.h:29-45: six divisions, glm min/max nesting, tEnter clamped at 0.
KK_HD void HitAABB(const Ray &r, const AABB &b, float &tEnter, float &tExit) {
    float h0 = (b.minPos.x - r.pos.x) / r.dir.x;
    float h1 = (b.maxPos.x - r.pos.x) / r.dir.x;
    float h2 = (b.minPos.y - r.pos.y) / r.dir.y;
    float h3 = (b.maxPos.y - r.pos.y) / r.dir.y;
    float h4 = (b.minPos.z - r.pos.z) / r.dir.z;
    float h5 = (b.maxPos.z - r.pos.z) / r.dir.z;
    tEnter = gmax(gmax(gmin(h0, h1), gmin(h2, h3)), gmin(h4, h5));
    tExit = gmin(gmin(gmax(h0, h1), gmax(h2, h3)), gmax(h4, h5));
    tEnter = (tEnter < 0.f) ? 0.f : tEnter;
}

// include/cg/ray.h:52-58 -> 3rd/glm/gtx/intersect.inl:28-94 (two-sided Moeller-Trumbore).
// Returns t = -1 on a miss.
KK_HD float HitTriangle(const Ray &r, f3 v0, f3 v1, f3 v2, float &bx, float &by) {
    f3 e1 = v1 - v0;
    f3 e2 = v2 - v0;
    f3 p = cross(r.dir, e2);
    float det = dot(e1, p);
    f3 perp;
    if (det > FloatEps) {
        f3 dist = r.pos - v0;
        bx = dot(dist, p);
        if (bx < 0.f || bx > det)
            return -1.f;
        perp = cross(dist, e1);
        by = dot(r.dir, perp);
        if (by < 0.f || (bx + by) > det)
            return -1.f;
    } else if (det < -FloatEps) {
        f3 dist = r.pos - v0;
        bx = dot(dist, p);
        if (bx > 0.f || bx < det)
            return -1.f;
        perp = cross(dist, e1);
        by = dot(r.dir, perp);
        if (by > 0.f || (bx + by) < det)
            return -1.f;
    } else
        return -1.f;
    float inv = 1.f / det;
    float t = dot(e2, perp) * inv;
    bx *= inv;
    by *= inv;
    return t;
}

// The same test with the same operations on the same operands, written without early exits: both signs
// of the determinant share one evaluation of dist, u, perp and v. For code in which the lanes of a wave
// test different triangles, where the branches above would each run for some lane.
KK_HD float HitTriangleFlat(const Ray &r, f3 v0, f3 v1, f3 v2, float &bx, float &by) {
    const f3 e1 = v1 - v0;
    const f3 e2 = v2 - v0;
    const f3 p = cross(r.dir, e2);
    const float det = dot(e1, p);
    const f3 dist = r.pos - v0;
    const float u = dot(dist, p);
    const f3 perp = cross(dist, e1);
    const float v = dot(r.dir, perp);
    const float uv = u + v;
    const bool front = det > FloatEps, back = det < -FloatEps;
    const bool inFront = !(u < 0.f || u > det) && !(v < 0.f || uv > det);
    const bool inBack = !(u > 0.f || u < det) && !(v > 0.f || uv < det);
    if (!((front && inFront) || (back && inBack)))
        return -1.f;
    const float inv = 1.f / det;
    const float t = dot(e2, perp) * inv;
    bx = u * inv;
    by = v * inv;
    return t;
}

} // namespace kk
