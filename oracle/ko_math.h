// TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the shipped library; it is the CPU
// checker that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg compare against.
//
// Small fp32 vector/matrix helpers that restate the operation ORDER of the vendored glm 0.9.9.8 the
// reference computes with, so that results can agree bit for bit when built with -ffp-contract=off:
//   min/max   : 3rd/glm/detail/func_common.inl:17-30
//   dot/cross : 3rd/glm/detail/func_geometric.inl:53, 75-77
//   length    : 3rd/glm/detail/func_geometric.inl:12
//   normalize : 3rd/glm/detail/func_geometric.inl:88 with inversesqrt = 1/sqrt (func_exponential.inl:136)
//   mat4*vec4 : 3rd/glm/detail/type_mat4x4.inl:563-571
//   mat3*vec3 : 3rd/glm/detail/type_mat3x3.inl:468-474
// tests/test_oracle_vs_glm.py checks every one of them against the real glm compiled into oracle/_ref.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

namespace ko {

constexpr float FloatMax = std::numeric_limits<float>::max();
constexpr float FloatMin = std::numeric_limits<float>::lowest();
constexpr float FloatEps = std::numeric_limits<float>::epsilon();

inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

struct vec2 {
    float x, y;
};

struct vec3 {
    float x, y, z;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
struct ivec3 {
    int32_t x, y, z;
    int32_t &operator[](int i) { return (&x)[i]; }
    const int32_t &operator[](int i) const { return (&x)[i]; }
};
struct uvec3 {
    uint32_t x, y, z;
    uint32_t &operator[](int i) { return (&x)[i]; }
    const uint32_t &operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
    float x, y, z, w;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};

inline vec3 V3(float a) { return {a, a, a}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(float s, vec3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(vec3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 &operator+=(vec3 &a, vec3 b) { return a = a + b; }
inline vec3 &operator*=(vec3 &a, vec3 b) { return a = a * b; }
inline bool operator==(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(vec3 a, vec3 b) { return !(a == b); }

inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator*(float s, vec4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator/(vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }

inline float dot(vec3 a, vec3 b) {
    vec3 t = a * b;
    return t.x + t.y + t.z;
}
inline vec3 cross(vec3 x, vec3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline float inversesqrt(float x) { return 1.f / std::sqrt(x); }
inline vec3 normalize(vec3 v) { return v * inversesqrt(dot(v, v)); }
inline vec3 vabs(vec3 v) { return {std::fabs(v.x), std::fabs(v.y), std::fabs(v.z)}; }
inline vec3 vfloor(vec3 v) { return {std::floor(v.x), std::floor(v.y), std::floor(v.z)}; }

// Column-major like glm: m.c[col] is a column vector.
struct mat3 {
    vec3 c[3];
};
struct mat4 {
    vec4 c[4];
};

inline vec3 operator*(const mat3 &m, vec3 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
inline vec4 operator*(const mat4 &m, vec4 v) {
    vec4 mul0 = m.c[0] * v.x;
    vec4 mul1 = m.c[1] * v.y;
    vec4 add0 = mul0 + mul1;
    vec4 mul2 = m.c[2] * v.z;
    vec4 mul3 = m.c[3] * v.w;
    vec4 add1 = mul2 + mul3;
    return add0 + add1;
}

// include/cg/bounding_box.h:10-66
struct AABB {
    vec3 minPos, maxPos;
    static AABB CreateInitial() { return {V3(FloatMax), V3(FloatMin)}; }
    void Expand(vec3 p) {
        for (int k = 0; k < 3; ++k) {
            if (minPos[k] > p[k])
                minPos[k] = p[k];
            if (maxPos[k] < p[k])
                maxPos[k] = p[k];
        }
    }
    void Expand(const AABB &o) {
        for (int k = 0; k < 3; ++k) {
            if (minPos[k] > o.minPos[k])
                minPos[k] = o.minPos[k];
            if (maxPos[k] < o.maxPos[k])
                maxPos[k] = o.maxPos[k];
        }
    }
};

struct Ray {
    vec3 pos, dir;
};

struct HitShell {
    float tEnter, tExit;
};

// include/cg/ray.h:29-45 (six divisions, glm min/max nesting, tEnter clamped at 0)
inline HitShell HitAABB(const Ray &r, const AABB &b) {
    float ht[6];
    ht[0] = (b.minPos.x - r.pos.x) / r.dir.x;
    ht[1] = (b.maxPos.x - r.pos.x) / r.dir.x;
    ht[2] = (b.minPos.y - r.pos.y) / r.dir.y;
    ht[3] = (b.maxPos.y - r.pos.y) / r.dir.y;
    ht[4] = (b.minPos.z - r.pos.z) / r.dir.z;
    ht[5] = (b.maxPos.z - r.pos.z) / r.dir.z;
    HitShell h;
    h.tEnter = gmax(gmax(gmin(ht[0], ht[1]), gmin(ht[2], ht[3])), gmin(ht[4], ht[5]));
    h.tExit = gmin(gmin(gmax(ht[0], ht[1]), gmax(ht[2], ht[3])), gmax(ht[4], ht[5]));
    h.tEnter = (h.tEnter < 0.f) ? 0.f : h.tEnter;
    return h;
}

struct HitTri {
    float t;
    vec2 bary;
};

// include/cg/ray.h:52-58 -> 3rd/glm/gtx/intersect.inl:28-94 (two-sided Moeller-Trumbore, miss -> t = -1)
inline HitTri HitTriangle(const Ray &r, vec3 v0, vec3 v1, vec3 v2) {
    HitTri ret;
    ret.bary = {0.f, 0.f};
    ret.t = -1.f;
    vec3 e1 = v1 - v0;
    vec3 e2 = v2 - v0;
    vec3 p = cross(r.dir, e2);
    float det = dot(e1, p);
    vec3 perp = {0.f, 0.f, 0.f};
    float bx, by;
    if (det > FloatEps) {
        vec3 dist = r.pos - v0;
        bx = dot(dist, p);
        if (bx < 0.f || bx > det)
            return ret;
        perp = cross(dist, e1);
        by = dot(r.dir, perp);
        if (by < 0.f || (bx + by) > det)
            return ret;
    } else if (det < -FloatEps) {
        vec3 dist = r.pos - v0;
        bx = dot(dist, p);
        if (bx > 0.f || bx < det)
            return ret;
        perp = cross(dist, e1);
        by = dot(r.dir, perp);
        if (by > 0.f || (bx + by) < det)
            return ret;
    } else
        return ret;
    float inv = 1.f / det;
    ret.t = dot(e2, perp) * inv;
    ret.bary = {bx * inv, by * inv};
    return ret;
}

} // namespace ko
