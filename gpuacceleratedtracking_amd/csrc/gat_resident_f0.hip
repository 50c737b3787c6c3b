// gat_resident_f0.hip -- instances of the resident correlator (gat_resident.h) for sample format GAT_LAYOUT_PLANAR, and the
// launcher over all four formats (one translation unit each, compiled in parallel).
#include "gat_resident.h"

namespace gat {
template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_PLANAR>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t);
extern template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t);
extern template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I16>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t);
extern template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I8>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t);

bool dc_has_resident_instance(int ant_tile, int taps, int format)
{
    return format >= GAT_LAYOUT_PLANAR && format <= GAT_LAYOUT_INTERLEAVED_I8 && dc_resident_instance(ant_tile, taps);
}

hipError_t launch_dc_resident(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s)
{
    if (!dc_has_resident_instance(cfg.ant_tile, cfg.taps, cfg.format)) return hipErrorInvalidValue;
    switch (cfg.format) {
    case GAT_LAYOUT_PLANAR: return launch_dc_resident_fmt<GAT_LAYOUT_PLANAR>(a, cfg, r, s);
    case GAT_LAYOUT_INTERLEAVED: return launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED>(a, cfg, r, s);
    case GAT_LAYOUT_INTERLEAVED_I16: return launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I16>(a, cfg, r, s);
    default: return launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I8>(a, cfg, r, s);
    }
}
} // namespace gat
