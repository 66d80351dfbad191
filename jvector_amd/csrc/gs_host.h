// gs_host.h — host-side preparation for the device-resident graph search (pure C++, no HIP): the open-addressing
// node -> row maps of the sparse upper levels that gs_body.h's gs_level_row probes, and the sizing rules.
#pragma once

#include <cstdint>
#include <vector>

namespace jv {

struct GsLevelMap {
    std::vector<int32_t> keys, vals;  // keys: node id or -1
    uint32_t mask = 0;
    int32_t shift = 32;
};

// nodes: the level's node ids (any order, unique).  Load factor <= 1/2, same hash as the device probe.
inline GsLevelMap gs_build_level_map(const int32_t *nodes, int count)
{
    GsLevelMap m;
    int log2 = 4;
    while ((1ll << log2) < 2ll * count) ++log2;
    const uint32_t size = 1u << log2;
    m.mask = size - 1;
    m.shift = 32 - log2;
    m.keys.assign(size, -1);
    m.vals.assign(size, -1);
    for (int i = 0; i < count; ++i) {
        uint32_t h = ((uint32_t)nodes[i] * 0x9E3779B1u) >> m.shift;
        while (m.keys[h] != -1) h = (h + 1) & m.mask;
        m.keys[h] = nodes[i];
        m.vals[h] = i;
    }
    return m;
}

// visited-table size for a search that keeps rerankK results: a search marks ~18 x rerankK nodes at maxDegree 32
// (measured, 10M x 768); 64 x leaves the table under half full for all but pathological queries (those overflow
// and are re-run on the host).
inline int gs_vcap_log2(int rerankK)
{
    int log2 = 12;
    while ((1ll << log2) < 64ll * rerankK && log2 < 22) ++log2;
    return log2;
}

}  // namespace jv
