// TEST INFRASTRUCTURE ONLY (see ko_math.h). CPU restatement of DepthBoxVDB::BuildFrom
// (src/ray_caster/dpbx_vdb.cu:275-610) and updateAtlas (:117-273).
//
// Parity status: the reference ships no golden vectors for this path and none of its .cu files builds
// here; this restatement is pinned by the structural invariants of
// include/ray_caster/test_options.h:22-78 (every node and leaf reached exactly once from the root, the
// atlas reverse map names the right leaf) and by parameter values derived by hand from the source for
// the two README data-set shapes (tests/test_oracle_vdb.py). "parity unpinned" beyond that.
//
// Point sampling of the volume (tex3D with cudaFilterModePoint, non-normalised coordinates, border
// addressing: dvr_dpbx_vdb.cpp:231-241) is "the voxel at floor(coordinate), 0 outside the volume".
#include "ko_vdb.h"

#include <algorithm>
#include <cstring>

using namespace ko;

namespace {

template <typename T> struct Volume {
    const T *dat;
    int dx, dy, dz;
    T at(int x, int y, int z) const {
        if (x < 0 || y < 0 || z < 0 || x >= dx || y >= dy || z >= dz)
            return 0;
        return dat[(static_cast<size_t>(z) * dy + y) * dx + x];
    }
    T atf(float x, float y, float z) const {
        return at(static_cast<int>(std::floor(x)), static_cast<int>(std::floor(y)),
                  static_cast<int>(std::floor(z)));
    }
};

// dpbx_vdb.cu:33-55
struct LevPos {
    uint8_t lev;
    int32_t pos[3];
    bool operator<(const LevPos &o) const {
        if (lev == o.lev) {
            if (pos[0] == o.pos[0]) {
                if (pos[1] == o.pos[1])
                    return pos[2] < o.pos[2];
                return pos[1] < o.pos[1];
            }
            return pos[0] < o.pos[0];
        }
        return lev < o.lev;
    }
    bool operator==(const LevPos &o) const {
        return lev == o.lev && pos[0] == o.pos[0] && pos[1] == o.pos[1] && pos[2] == o.pos[2];
    }
};

template <typename T> bool isEmpty(T v, const T *rngs, uint32_t n) {
    bool empty = false;
    for (uint32_t ri = 0; ri < n; ++ri)
        empty |= v >= rngs[2 * ri] && v <= rngs[2 * ri + 1];
    return empty;
}

// dpbx_vdb.cu:117-273 for the bricks that hold a leaf (tail bricks of the reference's atlas read
// unrelated nodes through the index formula at :191-194; they are not produced here).
template <typename T>
void updateAtlas(VDB<T> &vdb, const Volume<T> &vol, const T *rngs, uint32_t rngNum) {
    const VDBParameter &p = vdb.param;
    const int A = p.voxPerAtlasBrick;
    const size_t brickVox = static_cast<size_t>(A) * A * A;
    vdb.atlas.assign(brickVox * vdb.leafNum, 0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < static_cast<int64_t>(vdb.leafNum); ++k) {
        const Node &leaf = vdb.GetNode(0, static_cast<uint32_t>(k));
        T *out = vdb.atlas.data() + brickVox * k;
        for (int az = 0; az < A; ++az)
            for (int ay = 0; ay < A; ++ay)
                for (int ax = 0; ax < A; ++ax) {
                    const int a[3] = {ax, ay, az};
                    int depDir[3] = {0, 0, 0};
                    if (p.apronDepWid != p.apronWid)
                        for (int xyz = 0; xyz < 3; ++xyz)
                            depDir[xyz] = a[xyz] == 0 ? 1 : a[xyz] == A - 1 ? -1 : 0;
                    T &dst = out[(static_cast<size_t>(az) * A + ay) * A + ax];
                    if (depDir[0] == 0 && depDir[1] == 0 && depDir[2] == 0) {
                        dst = vol.atf(static_cast<float>(leaf.pos[0]) + ax - p.apronDepWid,
                                      static_cast<float>(leaf.pos[1]) + ay - p.apronDepWid,
                                      static_cast<float>(leaf.pos[2]) + az - p.apronDepWid);
                        continue;
                    }
                    if (std::abs(depDir[0]) + std::abs(depDir[1]) + std::abs(depDir[2]) > 1) {
                        dst = 0;
                        continue;
                    }
                    vec3 dltDir0 = depDir[0] != 0   ? vec3{0.f, 1.f, 0.f}
                                   : depDir[1] != 0 ? vec3{0.f, 0.f, 1.f}
                                                    : vec3{1.f, 0.f, 0.f};
                    vec3 dltDir1 = depDir[0] != 0   ? vec3{0.f, 0.f, 1.f}
                                   : depDir[1] != 0 ? vec3{1.f, 0.f, 0.f}
                                                    : vec3{0.f, 1.f, 0.f};
                    vec3 voxInVDB =
                        vec3{static_cast<float>(leaf.pos[0]), static_cast<float>(leaf.pos[1]),
                             static_cast<float>(leaf.pos[2])} +
                        vec3{depDir[0] == 1    ? 0.f
                             : depDir[0] == -1 ? static_cast<float>(p.dims[0] - 1)
                                               : static_cast<float>(ax - p.apronDepWid),
                             depDir[1] == 1    ? 0.f
                             : depDir[1] == -1 ? static_cast<float>(p.dims[0] - 1)
                                               : static_cast<float>(ay - p.apronDepWid),
                             depDir[2] == 1    ? 0.f
                             : depDir[2] == -1 ? static_cast<float>(p.dims[0] - 1)
                                               : static_cast<float>(az - p.apronDepWid)};
                    int32_t dep;
                    for (dep = 0; dep < p.dims[0] - 1; ++dep) {
                        bool stop = false;
                        for (int t = 0; t < 9; ++t) {
                            float dlt0 = (t % 3) - 1.f;
                            float dlt1 = (t / 3) - 1.f;
                            vec3 sp = voxInVDB + dlt0 * dltDir0 + dlt1 * dltDir1;
                            T v = vol.atf(sp.x, sp.y, sp.z);
                            if (!isEmpty(v, rngs, rngNum)) {
                                stop = true;
                                break;
                            }
                        }
                        if (stop)
                            break;
                        voxInVDB += vec3{static_cast<float>(depDir[0]), static_cast<float>(depDir[1]),
                                         static_cast<float>(depDir[2])};
                    }
                    dst = static_cast<T>(dep);
                }
    }
}

// dpbx_vdb.cu:275-610. Returns nullptr when the reference would print an error and stay incomplete
// (a log2 dimension outside [1, 10], an empty volume, three levels not covering the volume), and also
// when one brick covers the whole volume or no leaf is valid: the reference then leaves rootLev = 0 /
// builds empty pools, and its marcher indexes nodes[-1] (include/ray_caster/dda.h:23-27).
template <typename T>
VDB<T> *buildFrom(const T *volDat, int dx, int dy, int dz, const uint8_t log2Dims[3],
                  uint8_t apronWid, bool useDepthBox, const T *rngs, uint32_t rngNum) {
    for (int i = 0; i < 3; ++i)
        if (!(log2Dims[i] > 0 && log2Dims[i] <= 10))
            return nullptr;
    if (dx <= 0 || dy <= 0 || dz <= 0)
        return nullptr;
    Volume<T> vol{volDat, dx, dy, dz};
    auto *vdb = new VDB<T>();
    VDBParameter &p = vdb->param;
    std::memset(&p, 0, sizeof(p));
    p.apronWid = apronWid;
    p.apronDepWid = apronWid + (useDepthBox ? 1 : 0);
    for (int lev = 0; lev < MaxLevelNum; ++lev) {
        p.log2Dims[lev] = log2Dims[lev];
        p.dims[lev] = int32_t(1) << log2Dims[lev];
    }
    p.minDepPosValInBrick = -p.apronDepWid;
    p.maxDepPosValInBrick = p.dims[0] - 1 + p.apronDepWid;

    const uint32_t lx = (dx + p.dims[0] - 1) / p.dims[0];
    const uint32_t ly = (dy + p.dims[0] - 1) / p.dims[0];
    const uint32_t lz = (dz + p.dims[0] - 1) / p.dims[0];
    {
        p.rootLev = 0;
        p.voxsPerChild[0] = 1.f;
        uint64_t coverVox = p.dims[0];
        auto check = [&]() {
            return coverVox >= static_cast<uint64_t>(dx) && coverVox >= static_cast<uint64_t>(dy) &&
                   coverVox >= static_cast<uint64_t>(dz);
        };
        if (!check()) {
            int lev;
            for (lev = 1; lev < MaxLevelNum; ++lev) {
                p.voxsPerChild[lev] = static_cast<float>(coverVox);
                coverVox *= p.dims[lev];
                ++p.rootLev;
                if (check())
                    break;
            }
            if (lev == MaxLevelNum) {
                delete vdb;
                return nullptr;
            }
            p.voxPerVDB = static_cast<float>(coverVox);
            p.voxPerVol[0] = static_cast<float>(dx);
            p.voxPerVol[1] = static_cast<float>(dy);
            p.voxPerVol[2] = static_cast<float>(dz);
        }
    }
    if (p.rootLev == 0) {
        delete vdb;
        return nullptr;
    }

    // dpbx_vdb.cu:343-385: leaf validity (counts (voxel, range) pairs, as the reference does)
    const uint32_t lYxX = lx * ly;
    const uint32_t leafNum = lYxX * lz;
    std::vector<uint32_t> leafValids(leafNum, 1);
    const int B = p.dims[0];
    const uint32_t voxNumPerLeaf = static_cast<uint32_t>(B) * B * B;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t leafIdx = 0; leafIdx < static_cast<int64_t>(leafNum); ++leafIdx) {
        int pz = static_cast<int>(leafIdx / lYxX);
        int py = static_cast<int>((leafIdx - static_cast<int64_t>(pz) * lYxX) / lx);
        int px = static_cast<int>(leafIdx - static_cast<int64_t>(pz) * lYxX - static_cast<int64_t>(py) * lx);
        px *= B;
        py *= B;
        pz *= B;
        uint32_t emptyVoxNum = 0;
        for (int z = 0; z < B; ++z)
            for (int y = 0; y < B; ++y)
                for (int x = 0; x < B; ++x) {
                    T v = vol.at(px + x, py + y, pz + z);
                    for (uint32_t ri = 0; ri < rngNum; ++ri)
                        if (v >= rngs[2 * ri] && v <= rngs[2 * ri + 1])
                            ++emptyVoxNum;
                }
        if (emptyVoxNum == voxNumPerLeaf)
            leafValids[leafIdx] = 0;
    }
    std::vector<uint32_t> compactedLeaves;
    for (uint32_t i = 0; i < leafNum; ++i)
        if (leafValids[i])
            compactedLeaves.push_back(i);
    if (compactedLeaves.empty()) {
        delete vdb;
        return nullptr;
    }
    vdb->leafNum = static_cast<uint32_t>(compactedLeaves.size());

    // dpbx_vdb.cu:387-399
    p.voxPerAtlasBrick = p.dims[0] + p.apronDepWid * 2;
    p.atlasBrickPerVDB[0] = lx;
    p.atlasBrickPerVDB[1] = ly;
    p.atlasBrickPerVDB[2] = 1;
    while (static_cast<size_t>(p.atlasBrickPerVDB[0]) * p.atlasBrickPerVDB[1] * p.atlasBrickPerVDB[2] <
           compactedLeaves.size())
        ++p.atlasBrickPerVDB[2];
    vdb->atlasBrickToNodeIndices.assign(
        static_cast<size_t>(p.atlasBrickPerVDB[0]) * p.atlasBrickPerVDB[1] * p.atlasBrickPerVDB[2],
        UndefRltIdx);

    // dpbx_vdb.cu:404-445: (lev, pos) of every valid leaf and of its ancestors, sorted, unique
    std::vector<LevPos> levPoss;
    for (uint32_t c = 0; c < compactedLeaves.size(); ++c) {
        uint32_t leafIdx = compactedLeaves[c];
        LevPos lp;
        lp.lev = 0;
        lp.pos[2] = leafIdx / lYxX;
        lp.pos[1] = (leafIdx - lp.pos[2] * lYxX) / lx;
        lp.pos[0] = leafIdx - lp.pos[2] * lYxX - lp.pos[1] * lx;
        levPoss.push_back(lp);
        for (uint8_t lev = 1; lev <= p.rootLev; ++lev) {
            for (int k = 0; k < 3; ++k)
                lp.pos[k] = lp.pos[k] / p.dims[lev];
            lp.lev = lev;
            levPoss.push_back(lp);
        }
    }
    std::sort(levPoss.begin(), levPoss.end());
    levPoss.erase(std::unique(levPoss.begin(), levPoss.end()), levPoss.end());

    // dpbx_vdb.cu:447-467
    uint32_t nodeNums[MaxLevelNum] = {0, 0, 0};
    uint64_t totNodeNum = 0, totChildNum = 0;
    {
        size_t cur = 0;
        for (uint8_t lev = 0; lev <= p.rootLev; ++lev) {
            size_t nxt = cur;
            while (nxt < levPoss.size() && levPoss[nxt].lev == lev)
                ++nxt;
            nodeNums[lev] = static_cast<uint32_t>(nxt - cur);
            p.nodeStarts[lev] = totNodeNum;
            totNodeNum += nodeNums[lev];
            cur = nxt;
            if (lev != 0) {
                uint64_t childNum = static_cast<uint64_t>(nodeNums[lev]) * p.dims[lev] * p.dims[lev] * p.dims[lev];
                p.childStarts[lev - 1] = totChildNum;
                totChildNum += childNum;
            }
        }
    }

    // dpbx_vdb.cu:57-86, 524-544: nodes
    vdb->nodePools.assign(totNodeNum, Node{});
    for (uint8_t lev = 0; lev < p.rootLev; ++lev)
        for (uint32_t k = 0; k < nodeNums[lev]; ++k) {
            Node inode{};
            std::memcpy(inode.pos, levPoss[p.nodeStarts[lev] + k].pos, 12);
            if (lev == 0) {
                inode.childPoolOffsRelative = UndefRltIdx;
                inode.brickPosInAtlas[2] = k / lYxX;
                inode.brickPosInAtlas[1] = (k - inode.brickPosInAtlas[2] * lYxX) / lx;
                inode.brickPosInAtlas[0] =
                    k - inode.brickPosInAtlas[1] * lx - inode.brickPosInAtlas[2] * lYxX;
                vdb->atlasBrickToNodeIndices[static_cast<size_t>(inode.brickPosInAtlas[2]) *
                                                 p.atlasBrickPerVDB[1] * p.atlasBrickPerVDB[0] +
                                             inode.brickPosInAtlas[1] * p.atlasBrickPerVDB[0] +
                                             inode.brickPosInAtlas[0]] = k;
            } else
                inode.childPoolOffsRelative =
                    static_cast<uint64_t>(k) * p.dims[lev] * p.dims[lev] * p.dims[lev];
            vdb->GetNode(lev, k) = inode;
        }
    {
        Node root{};
        root.childPoolOffsRelative = 0;
        vdb->nodePools[totNodeNum - 1] = root;
    }

    // dpbx_vdb.cu:88-115, 546-558: children, top level first
    vdb->childPools.assign(totChildNum, UndefRltIdx);
    for (uint8_t compoLev = 0; compoLev < p.rootLev; ++compoLev) {
        uint8_t lev = p.rootLev - 1 - compoLev;
        for (uint32_t k = 0; k < nodeNums[lev]; ++k) {
            Node node = vdb->GetNode(lev, k);
            uint8_t parLev = p.rootLev;
            Node par = vdb->nodePools[p.nodeStarts[p.rootLev]];
            while (parLev > lev + 1) {
                int32_t cover = 1;
                for (int upLev = lev + 1; upLev < p.rootLev; ++upLev)
                    cover *= p.dims[upLev];
                int32_t pos[3] = {node.pos[0] / cover, node.pos[1] / cover, node.pos[2] / cover};
                par = vdb->GetNode(parLev - 1, vdb->GetChild(parLev, par.childPoolOffsRelative, pos));
                --parLev;
            }
            int32_t rel[3];
            for (int q = 0; q < 3; ++q)
                rel[q] = node.pos[q] - par.pos[q] * p.dims[parLev];
            vdb->childPools[p.childStarts[parLev - 1] + par.childPoolOffsRelative +
                            vdb->PosRelativeToIdxRelative(parLev, rel)] = k;
        }
    }

    // dpbx_vdb.cu:560-571: level units -> voxels
    for (uint8_t lev = 0; lev < p.rootLev; ++lev)
        for (uint32_t k = 0; k < nodeNums[lev]; ++k) {
            Node &n = vdb->GetNode(lev, k);
            for (int l = lev + 1; l > 0; --l)
                for (int q = 0; q < 3; ++q)
                    n.pos[q] *= p.dims[l - 1];
        }

    updateAtlas(*vdb, vol, rngs, rngNum);
    return vdb;
}

template <typename T> void sizesOf(const VDB<T> *v, uint64_t *out) {
    out[0] = v->nodePools.size();
    out[1] = v->childPools.size();
    out[2] = v->atlasBrickToNodeIndices.size();
    out[3] = v->atlas.size();
    out[4] = v->leafNum;
}
template <typename T> void getFrom(const VDB<T> *v, int what, void *dst) {
    switch (what) {
    case 0:
        std::memcpy(dst, &v->param, sizeof(VDBParameter));
        break;
    case 1:
        std::memcpy(dst, v->nodePools.data(), v->nodePools.size() * sizeof(Node));
        break;
    case 2:
        std::memcpy(dst, v->childPools.data(), v->childPools.size() * 4);
        break;
    case 3:
        std::memcpy(dst, v->atlasBrickToNodeIndices.data(), v->atlasBrickToNodeIndices.size() * 4);
        break;
    case 4:
        std::memcpy(dst, v->atlas.data(), v->atlas.size() * sizeof(T));
        break;
    }
}

} // namespace

extern "C" {

void *ko_vdb_build_u8(const uint8_t *vol, int dx, int dy, int dz, const uint8_t *log2Dims,
                      uint8_t apronWid, int useDepthBox, const uint8_t *rngs, uint32_t rngNum) {
    return buildFrom<uint8_t>(vol, dx, dy, dz, log2Dims, apronWid, useDepthBox != 0, rngs, rngNum);
}
void *ko_vdb_build_u16(const uint16_t *vol, int dx, int dy, int dz, const uint8_t *log2Dims,
                       uint8_t apronWid, int useDepthBox, const uint16_t *rngs, uint32_t rngNum) {
    return buildFrom<uint16_t>(vol, dx, dy, dz, log2Dims, apronWid, useDepthBox != 0, rngs, rngNum);
}
void ko_vdb_sizes(const void *h, int is16, uint64_t *out5) {
    if (is16)
        sizesOf(static_cast<const VDB<uint16_t> *>(h), out5);
    else
        sizesOf(static_cast<const VDB<uint8_t> *>(h), out5);
}
// what: 0 VDBParameter (128 B), 1 nodes, 2 child pools, 3 atlas-brick-to-node map, 4 atlas (brick-major)
void ko_vdb_get(const void *h, int is16, int what, void *dst) {
    if (is16)
        getFrom(static_cast<const VDB<uint16_t> *>(h), what, dst);
    else
        getFrom(static_cast<const VDB<uint8_t> *>(h), what, dst);
}
void ko_vdb_free(void *h, int is16) {
    if (is16)
        delete static_cast<VDB<uint16_t> *>(h);
    else
        delete static_cast<VDB<uint8_t> *>(h);
}

} // extern "C"
