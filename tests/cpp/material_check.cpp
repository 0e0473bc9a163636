// Evaluates the host-callable members of kouek::RayTracer::Material and Light (BRDF, SampleQuad,
// SampleSphere) for the inputs of a binary file and writes the results, for the parity test against
// the CPU checker (tests/test_oracle_pt.py). Plain g++: no GPU call is made.
//   in : uint32 n, then n records {Material 64 B, Light 80 B, float dirIn[3], dirOut[3], norm[3], u0, u1, pos[3]}
//   out: n records {float brdf[3], float samplePos[3], sampleNorm[3], samplePdf}
#include <cstdio>
#include <cstring>
#include <vector>

#include <kouek/ray_tracer/material.h>

using namespace kouek;
using namespace kouek::RayTracer;

int main(int argc, char **argv) {
    if (argc != 3)
        return 2;
    FILE *fi = std::fopen(argv[1], "rb"), *fo = std::fopen(argv[2], "wb");
    if (!fi || !fo)
        return 3;
    uint32_t n = 0;
    if (std::fread(&n, 4, 1, fi) != 1)
        return 4;
    for (uint32_t i = 0; i < n; ++i) {
        Material m;
        Light l;
        float f[14];
        if (std::fread(&m, sizeof(m), 1, fi) != 1 || std::fread(&l, sizeof(l), 1, fi) != 1 ||
            std::fread(f, 4, 14, fi) != 14)
            return 5;
        const vec3 brdf = m.BRDF(vec3(f[0], f[1], f[2]), vec3(f[3], f[4], f[5]), vec3(f[6], f[7], f[8]));
        const Light::SampleResult s = l.type == Light::Type::Quad
                                          ? l.SampleQuad(f[9], f[10])
                                          : l.SampleSphere(f[9], f[10], vec3(f[11], f[12], f[13]));
        const float out[10] = {brdf[0], brdf[1], brdf[2], s.pos[0],  s.pos[1],
                               s.pos[2], s.norm[0], s.norm[1], s.norm[2], s.pdf};
        std::fwrite(out, 4, 10, fo);
    }
    std::fclose(fi);
    std::fclose(fo);
    return 0;
}
