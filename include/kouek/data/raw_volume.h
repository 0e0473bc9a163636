// kouek::Data::RAWVolume<T> with the reference's interface (reference include/data/raw_volume.h:17-113):
// reads dim.x * dim.y * dim.z voxels of type T (x fastest) and applies the axis transform trAxis, a
// signed permutation of (1, 2, 3): output axis k takes input axis |trAxis[k]|, reversed when negative.
#ifndef KOUEK_DATA_RAW_VOLUME_H
#define KOUEK_DATA_RAW_VOLUME_H

#include <array>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <string_view>
#include <vector>

#include "../math_types.h"

namespace kouek {
namespace Data {

template <typename T> class RAWVolume {
    bool isComplete = false;
    uvec3 dim;
    std::vector<T> dat;
    static constexpr std::string_view ErrTag = "[RAWVolume Error]";

  public:
    RAWVolume(const std::string &path, const uvec3 &srcDim, const std::array<int8_t, 3> &trAxis = {1, 2, 3}) {
        int seen[3] = {0, 0, 0};
        bool valid = true;
        for (int k = 0; k < 3; ++k) {
            const int ax = std::abs(trAxis[k]);
            if (ax < 1 || ax > 3)
                valid = false;
            else
                ++seen[ax - 1];
        }
        if (!valid || seen[0] != 1 || seen[1] != 1 || seen[2] != 1) {
            std::cerr << ErrTag << " Parameter trAxis is invalid.\n";
            return;
        }
        std::ifstream is(path, std::ios::in | std::ios::binary | std::ios::ate);
        if (!is.is_open()) {
            std::cerr << ErrTag << " Cannot open file at \"" << path << "\".\n";
            return;
        }
        const size_t voxNum = static_cast<size_t>(srcDim[0]) * srcDim[1] * srcDim[2];
        if (static_cast<size_t>(is.tellg()) < voxNum * sizeof(T)) {
            std::cerr << ErrTag << " File " << path << " is too small to contain volume of size (" << srcDim[0]
                      << "," << srcDim[1] << "," << srcDim[2] << ")\n";
            return;
        }
        is.seekg(0);
        std::vector<T> src(voxNum);
        is.read(reinterpret_cast<char *>(src.data()), static_cast<std::streamsize>(voxNum * sizeof(T)));
        const int m[3] = {std::abs(trAxis[0]) - 1, std::abs(trAxis[1]) - 1, std::abs(trAxis[2]) - 1};
        dim = uvec3(srcDim[m[0]], srcDim[m[1]], srcDim[m[2]]);
        if (trAxis[0] == 1 && trAxis[1] == 2 && trAxis[2] == 3)
            dat = std::move(src);
        else {
            dat.resize(voxNum);
            size_t offs = 0;
            uint32_t c[3];
            for (c[2] = 0; c[2] < srcDim[2]; ++c[2])
                for (c[1] = 0; c[1] < srcDim[1]; ++c[1])
                    for (c[0] = 0; c[0] < srcDim[0]; ++c[0], ++offs) {
                        uint32_t t[3];
                        for (int k = 0; k < 3; ++k)
                            t[k] = trAxis[k] > 0 ? c[m[k]] : dim[k] - 1 - c[m[k]];
                        dat[(static_cast<size_t>(t[2]) * dim[1] + t[1]) * dim[0] + t[0]] = src[offs];
                    }
        }
        isComplete = true;
    }

    bool IsComplete() const { return isComplete; }
    const std::vector<T> &Get() const { return dat; }
    const uvec3 &GetDimension() const { return dim; }
};

} // namespace Data
} // namespace kouek

#endif // KOUEK_DATA_RAW_VOLUME_H
