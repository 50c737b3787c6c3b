// gat_resident_f0.hip -- instances of the resident correlator (gat_resident.h) for sample format GAT_LAYOUT_PLANAR, and the
// launcher over all four formats (one translation unit each, compiled in parallel).
#include "gat_resident.h"

namespace gat {
template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_PLANAR>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t, int *);
extern template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t, int *);
extern template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I16>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t, int *);
extern template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I8>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t, int *);

bool dc_has_resident_instance(int ant_tile, int taps, int format)
{
    return format >= GAT_LAYOUT_PLANAR && format <= GAT_LAYOUT_INTERLEAVED_I8 && dc_resident_instance(ant_tile, taps);
}

static hipError_t resident_dispatch(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s, int *blocks_per_cu)
{
    if (!dc_has_resident_instance(cfg.ant_tile, cfg.taps, cfg.format)) return hipErrorInvalidValue;
    switch (cfg.format) {
    case GAT_LAYOUT_PLANAR: return launch_dc_resident_fmt<GAT_LAYOUT_PLANAR>(a, cfg, r, s, blocks_per_cu);
    case GAT_LAYOUT_INTERLEAVED: return launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED>(a, cfg, r, s, blocks_per_cu);
    case GAT_LAYOUT_INTERLEAVED_I16: return launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I16>(a, cfg, r, s, blocks_per_cu);
    default: return launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I8>(a, cfg, r, s, blocks_per_cu);
    }
}

hipError_t launch_dc_resident(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s)
{
    return resident_dispatch(a, cfg, r, s, nullptr);
}

hipError_t dc_resident_blocks_per_cu(const DcLaunch &cfg, int *blocks_per_cu)
{
    if (!blocks_per_cu) return hipErrorInvalidValue;
    *blocks_per_cu = 0;
    return resident_dispatch(DcArgs{}, cfg, ResidentArgs{}, nullptr, blocks_per_cu);
}
} // namespace gat
