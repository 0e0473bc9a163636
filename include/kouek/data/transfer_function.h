// kouek::Data::TransferFunction<T, Min, Max, IncStep> with the reference's interface and rules
// (reference include/data/transfer_function.h:18-127): control points "scalar r g b a" (colours over
// 255) from a text file -> a table of Max - Min + 1 RGBA entries by piecewise-linear interpolation, and
// the scalar ranges in which the opacity is zero. The reference's behaviours are kept: the
// interpolation cursor moves on by at most one control point per scalar, the last entry is the cursor's
// point, and an empty range is emitted only once a non-empty stretch follows it; ranges are stored
// truncated to T.
#ifndef KOUEK_DATA_TRANSFER_FUNCTION_H
#define KOUEK_DATA_TRANSFER_FUNCTION_H

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <string_view>
#include <vector>

#include "../math_types.h"

namespace kouek {
namespace Data {

template <typename T, T Min, T Max, T IncStep = 1> class TransferFunction {
  public:
    static constexpr auto MaxScalar = Max;
    static constexpr auto MinScalar = Min;

  private:
    bool isComplete = false;
    std::vector<vec4> dat;
    std::map<float, vec4> pnts;
    vec2 scalarRng;
    std::vector<vec2_of<T>> emptyScalarRngs;
    static constexpr std::string_view ErrTag = "[TransferFunction Error]";

  public:
    explicit TransferFunction(const std::string &path) {
        std::ifstream is(path);
        if (!is.is_open()) {
            std::cerr << ErrTag << " Cannot open file at \"" << path << "\".\n";
            return;
        }
        std::stringstream ss;
        ss << is.rdbuf();
        parse(ss.str());
    }
    // the same from the file's text (the README tables)
    static TransferFunction FromText(const std::string &text) {
        TransferFunction tf;
        tf.parse(text);
        return tf;
    }

    bool IsComplete() const { return isComplete; }
    const std::vector<vec4> &Get() const { return dat; }
    const std::map<float, vec4> &GetPoints() const { return pnts; }
    const vec2 &GetScalarRange() const { return scalarRng; }
    const std::vector<vec2_of<T>> &GetEmptyScalarRanges() const { return emptyScalarRngs; }

  private:
    TransferFunction() = default;

    void parse(const std::string &text) {
        scalarRng[0] = static_cast<float>(Max);
        scalarRng[1] = static_cast<float>(Min);
        std::istringstream lines(text);
        std::string line;
        uint32_t ln = 0;
        while (std::getline(lines, line)) {
            float v[5];
            const char *s = line.c_str();
            int got = 0;
            for (; got < 5; ++got) {
                char *end = nullptr;
                v[got] = std::strtof(s, &end);
                if (end == s)
                    break;
                s = end;
            }
            if (got != 5) {
                std::cerr << ErrTag << " Line " << ln << " is of invalid format.\n";
                return;
            }
            ++ln;
            if (v[0] < scalarRng[0])
                scalarRng[0] = v[0];
            if (v[0] > scalarRng[1])
                scalarRng[1] = v[0];
            vec4 col;
            for (int i = 0; i < 4; ++i)
                col[i] = v[1 + i] / 255.f;
            pnts.emplace(v[0], col);
        }
        if (scalarRng[0] != static_cast<float>(Min) || scalarRng[1] != static_cast<float>(Max)) {
            std::cerr << ErrTag << " The points do not span the scalar range.\n";
            return;
        }
        compute();
        isComplete = true;
    }

    void compute() {
        dat.clear();
        emptyScalarRngs.clear();
        auto itr = pnts.begin(), prev = pnts.begin();
        float lo = static_cast<float>(Max), hi = static_cast<float>(Min);
        for (T scalar = Min; scalar < Max; scalar += IncStep) {
            if (static_cast<float>(scalar) > itr->first) {
                prev = itr;
                ++itr;
            }
            const float s = itr->first, ps = prev->first;
            const vec4 &c = itr->second, &pc = prev->second;
            if (pc[3] == 0.f && c[3] == 0.f) {
                if (lo > ps)
                    lo = ps;
                if (hi < s)
                    hi = s;
            } else if (lo != static_cast<float>(Max) && hi != static_cast<float>(Min)) {
                vec2_of<T> r;
                r[0] = static_cast<T>(lo);
                r[1] = static_cast<T>(hi);
                emptyScalarRngs.push_back(r);
                lo = static_cast<float>(Max), hi = static_cast<float>(Min);
            }
            const float a = s == ps ? 1.f : (static_cast<float>(scalar) - ps) / (s - ps);
            vec4 e;
            for (int i = 0; i < 4; ++i)
                e[i] = (1.f - a) * pc[i] + a * c[i];
            dat.push_back(e);
        }
        dat.push_back(itr->second);
    }
};

} // namespace Data
} // namespace kouek

#endif // KOUEK_DATA_TRANSFER_FUNCTION_H
