// Vector / matrix types of the host-side classes. The reference's public API is written with glm
// (glm::vec3, glm::mat4, glm::vec<3, uint32_t> ...). A project that has glm defines KOUEK_WITH_GLM and
// gets exactly those types; otherwise the minimal layout-compatible types below are used (column-major
// matrices, m[col][row], like glm).
#ifndef KOUEK_MATH_TYPES_H
#define KOUEK_MATH_TYPES_H

#include <array>
#include <cmath>
#include <cstdint>

#ifdef KOUEK_WITH_GLM
#include <glm/glm.hpp>
namespace kouek {
using vec2 = glm::vec2;
using vec3 = glm::vec3;
using vec4 = glm::vec4;
using ivec2 = glm::ivec2;
using uvec3 = glm::vec<3, uint32_t>;
using mat3 = glm::mat3;
using mat4 = glm::mat4;
template <typename T> using vec2_of = glm::vec<2, T>;
} // namespace kouek
#else
namespace kouek {
struct vec2 {
    float x, y;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float a) : x(a), y(a), z(a) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
    bool operator==(const vec3 &o) const { return x == o.x && y == o.y && z == o.z; }
    bool operator!=(const vec3 &o) const { return !(*this == o); }
};
struct vec4 {
    float x, y, z, w;
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
struct ivec2 {
    int x, y;
    bool operator==(const ivec2 &o) const { return x == o.x && y == o.y; }
    bool operator!=(const ivec2 &o) const { return !(*this == o); }
};
struct uvec3 {
    uint32_t x, y, z;
    uvec3() : x(0), y(0), z(0) {}
    uvec3(uint32_t a, uint32_t b, uint32_t c) : x(a), y(b), z(c) {}
    uint32_t &operator[](int i) { return (&x)[i]; }
    const uint32_t &operator[](int i) const { return (&x)[i]; }
};
struct mat3 {
    vec3 c[3];
    vec3 &operator[](int i) { return c[i]; }
    const vec3 &operator[](int i) const { return c[i]; }
    bool operator!=(const mat3 &o) const { return c[0] != o.c[0] || c[1] != o.c[1] || c[2] != o.c[2]; }
};
struct mat4 {
    vec4 c[4];
    vec4 &operator[](int i) { return c[i]; }
    const vec4 &operator[](int i) const { return c[i]; }
    static mat4 identity() {
        mat4 m{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                m.c[i][j] = i == j ? 1.f : 0.f;
        return m;
    }
};
template <typename T> using vec2_of = std::array<T, 2>;
} // namespace kouek
#endif // KOUEK_WITH_GLM

namespace kouek {
namespace Math {

// kouek::Math::InverseProjective (reference include/cg/math.h:14-52): block-wise inverse of a
// projection matrix, identity when the Schur complement is singular. Written on raw column-major
// floats so that it is the same code for both sets of types.
inline mat4 InverseProjective(const mat4 &projection) {
    auto P = [&](int c, int r) { return projection[c][r]; };
    struct M2 {
        float v[2][2]; // [col][row]
    };
    auto inv2 = [](const M2 &m) {
        float d = 1.f / (m.v[0][0] * m.v[1][1] - m.v[1][0] * m.v[0][1]);
        return M2{{{m.v[1][1] * d, -m.v[0][1] * d}, {-m.v[1][0] * d, m.v[0][0] * d}}};
    };
    auto mul2 = [](const M2 &a, const M2 &b) {
        M2 r;
        for (int c = 0; c < 2; ++c)
            for (int row = 0; row < 2; ++row)
                r.v[c][row] = a.v[0][row] * b.v[c][0] + a.v[1][row] * b.v[c][1];
        return r;
    };
    auto sub2 = [](const M2 &a, const M2 &b) {
        M2 r;
        for (int c = 0; c < 2; ++c)
            for (int row = 0; row < 2; ++row)
                r.v[c][row] = a.v[c][row] - b.v[c][row];
        return r;
    };
    M2 a{{{P(0, 0), P(0, 1)}, {P(1, 0), P(1, 1)}}};
    M2 b{{{P(2, 0), P(2, 1)}, {P(3, 0), P(3, 1)}}};
    M2 c{{{P(0, 2), P(0, 3)}, {P(1, 2), P(1, 3)}}};
    M2 d{{{P(2, 2), P(2, 3)}, {P(3, 2), P(3, 3)}}};
    a = inv2(a);
    M2 ab = mul2(a, b);
    M2 ca = mul2(c, a);
    M2 cab = mul2(ca, b);
    M2 dcab = sub2(d, cab);
    float det = dcab.v[1][1] * dcab.v[0][0] - dcab.v[1][0] * dcab.v[0][1];
    mat4 out{};
    if (std::fabs(det) <= 1.192092896e-07f) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                out[i][j] = i == j ? 1.f : 0.f;
        return out;
    }
    M2 d1 = inv2(dcab);
    M2 d2{{{-d1.v[0][0], -d1.v[0][1]}, {-d1.v[1][0], -d1.v[1][1]}}};
    M2 c1 = mul2(d2, ca);
    M2 b1 = mul2(ab, d2);
    M2 a1 = sub2(a, mul2(ab, c1));
    const float e[16] = {a1.v[0][0], a1.v[0][1], c1.v[0][0], c1.v[0][1], a1.v[1][0], a1.v[1][1],
                         c1.v[1][0], c1.v[1][1], b1.v[0][0], b1.v[0][1], d1.v[0][0], d1.v[0][1],
                         b1.v[1][0], b1.v[1][1], d1.v[1][0], d1.v[1][1]};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            out[i][j] = e[4 * i + j];
    return out;
}

} // namespace Math
} // namespace kouek

#endif // KOUEK_MATH_TYPES_H
