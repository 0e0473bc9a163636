// TEST INFRASTRUCTURE ONLY. Thin extern "C" shim over the parts of the reference that compile
// unmodified with g++ in this image: the vendored glm 0.9.9.8 (3rd/glm) and include/cg/FPS_camera.h.
// Built by oracle/Makefile straight from /root/reference into oracle/_ref/libkkref.so; no reference
// source is copied. Everything else in the reference needs nvcc / CUDA headers / <format> and is
// treated as unbuildable here.
#define GLM_ENABLE_EXPERIMENTAL
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtx/intersect.hpp>

#include <cg/FPS_camera.h>

#include <cstring>

extern "C" {

// glm::intersectRayTriangle as called by include/cg/ray.h:52-58
void kkref_hit_triangle(const float *ray6, const float *tri9, float *out3) {
    glm::vec3 pos(ray6[0], ray6[1], ray6[2]), dir(ray6[3], ray6[4], ray6[5]);
    glm::vec3 p0(tri9[0], tri9[1], tri9[2]), p1(tri9[3], tri9[4], tri9[5]),
        p2(tri9[6], tri9[7], tri9[8]);
    glm::vec2 bary(0.f);
    float t = -1.f;
    if (!glm::intersectRayTriangle(pos, dir, p0, p1, p2, bary, t)) {
        t = -1.f;
        bary = glm::vec2(0.f);
    }
    out3[0] = t;
    out3[1] = bary.x;
    out3[2] = bary.y;
}

// The expression of include/cg/ray.h:33-43 written with the real glm::min / glm::max.
void kkref_hit_aabb(const float *ray6, const float *aabb6, float *out2) {
    glm::vec3 pos(ray6[0], ray6[1], ray6[2]), dir(ray6[3], ray6[4], ray6[5]);
    float ht[6];
    ht[0] = (aabb6[0] - pos.x) / dir.x;
    ht[1] = (aabb6[3] - pos.x) / dir.x;
    ht[2] = (aabb6[1] - pos.y) / dir.y;
    ht[3] = (aabb6[4] - pos.y) / dir.y;
    ht[4] = (aabb6[2] - pos.z) / dir.z;
    ht[5] = (aabb6[5] - pos.z) / dir.z;
    float tEnter =
        glm::max(glm::max(glm::min(ht[0], ht[1]), glm::min(ht[2], ht[3])), glm::min(ht[4], ht[5]));
    float tExit =
        glm::min(glm::min(glm::max(ht[0], ht[1]), glm::max(ht[2], ht[3])), glm::max(ht[4], ht[5]));
    out2[0] = (tEnter < 0.f) ? 0.f : tEnter;
    out2[1] = tExit;
}

void kkref_normalize(const float *v3, float *out3) {
    glm::vec3 n = glm::normalize(glm::vec3(v3[0], v3[1], v3[2]));
    std::memcpy(out3, &n, 12);
}
void kkref_cross(const float *a3, const float *b3, float *out3) {
    glm::vec3 n = glm::cross(glm::vec3(a3[0], a3[1], a3[2]), glm::vec3(b3[0], b3[1], b3[2]));
    std::memcpy(out3, &n, 12);
}
float kkref_dot(const float *a3, const float *b3) {
    return glm::dot(glm::vec3(a3[0], a3[1], a3[2]), glm::vec3(b3[0], b3[1], b3[2]));
}
void kkref_mat4_mul_vec4(const float *m16, const float *v4, float *out4) {
    glm::mat4 m;
    std::memcpy(&m, m16, 64);
    glm::vec4 r = m * glm::vec4(v4[0], v4[1], v4[2], v4[3]);
    std::memcpy(out4, &r, 16);
}
void kkref_mat3_mul_vec3(const float *m9, const float *v3, float *out3) {
    glm::mat3 m;
    std::memcpy(&m, m9, 36);
    glm::vec3 r = m * glm::vec3(v3[0], v3[1], v3[2]);
    std::memcpy(out3, &r, 12);
}
void kkref_mat4_mul_mat4(const float *a16, const float *b16, float *out16) {
    glm::mat4 a, b;
    std::memcpy(&a, a16, 64);
    std::memcpy(&b, b16, 64);
    glm::mat4 r = a * b;
    std::memcpy(out16, &r, 64);
}
// app/rt_lbvh.cpp:239-241, app/dvr_dpbx_vdb.cpp:322-324
void kkref_perspective_fov(float fovDeg, float w, float h, float n, float f, float *out16) {
    glm::mat4 p = glm::perspectiveFov(glm::radians(fovDeg), w, h, n, f);
    std::memcpy(out16, &p, 64);
}
// glm::inverse(mat2) and the mat2 products include/cg/math.h:14-52 is written with
void kkref_mat2_inverse(const float *m4, float *out4) {
    glm::mat2 m;
    std::memcpy(&m, m4, 16);
    glm::mat2 r = glm::inverse(m);
    std::memcpy(out4, &r, 16);
}
void kkref_mat2_mul(const float *a4, const float *b4, float *out4) {
    glm::mat2 a, b;
    std::memcpy(&a, a4, 16);
    std::memcpy(&b, b4, 16);
    glm::mat2 r = a * b;
    std::memcpy(out4, &r, 16);
}
// app/dvr_dpbx_vdb.cpp:272-273
void kkref_w2s_from_scales(const float *scales3, float *out16) {
    glm::vec3 scales(scales3[0], scales3[1], scales3[2]);
    glm::mat4 w2s = glm::scale(glm::identity<glm::mat4>(), 1.f / scales) *
                    glm::translate(glm::identity<glm::mat4>(), glm::vec3(.5f) * scales);
    std::memcpy(out16, &w2s, 64);
}

// include/cg/FPS_camera.h: LookAt then `steps` calls of Revolve(dist(pos, 0), yawDeg, 0) as in the
// perf orbit of app/dvr_dpbx_vdb.cpp:352-367. out12 = right, forward, up, pos.
void kkref_camera_orbit(const float *eye3, const float *center3, const float *up3, int steps,
                        float yawDeg, float *out12) {
    kouek::FPSCamera cam;
    cam.LookAt(glm::vec3(eye3[0], eye3[1], eye3[2]), glm::vec3(center3[0], center3[1], center3[2]),
               glm::vec3(up3[0], up3[1], up3[2]));
    for (int i = 0; i < steps; ++i)
        cam.Revolve(glm::distance(cam.GetPos(), glm::zero<glm::vec3>()), yawDeg, 0.f);
    const auto &[R, F, U, P] = cam.GetRFUP();
    std::memcpy(out12 + 0, &R, 12);
    std::memcpy(out12 + 3, &F, 12);
    std::memcpy(out12 + 6, &U, 12);
    std::memcpy(out12 + 9, &P, 12);
}

} // extern "C"
