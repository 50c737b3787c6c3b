// gat_resident_f0.hip -- instances of the resident correlator (gat_resident.h) for sample format GAT_LAYOUT_PLANAR.
#include "gat_resident.h"

namespace gat {
template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_PLANAR>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t);
extern template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t);

bool dc_has_resident_instance(int ant_tile, int taps, int format)
{
    return (format == GAT_LAYOUT_PLANAR || format == GAT_LAYOUT_INTERLEAVED) && dc_resident_instance(ant_tile, taps);
}

hipError_t launch_dc_resident(const DcArgs &a, const DcLaunch &cfg, const ResidentArgs &r, hipStream_t s)
{
    if (!dc_has_resident_instance(cfg.ant_tile, cfg.taps, cfg.format)) return hipErrorInvalidValue;
    if (cfg.format == GAT_LAYOUT_PLANAR) return launch_dc_resident_fmt<GAT_LAYOUT_PLANAR>(a, cfg, r, s);
    return launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED>(a, cfg, r, s);
}
} // namespace gat
