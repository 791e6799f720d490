// rgcn_tile_fp32_wide.hip -- the instantiations of rgcn_tile_kernel for gathered widths 64 and 128 (rgcn_tile_fp32_kernel.h).
#include "rgcn_tile_fp32_kernel.h"

namespace rgcn {

int dispatch_tile_wide(int KP, int NP, const TileArgs& a, int n_tiles, int chunk, void* s) {
    switch (KP) {
        case 64: return dispatch_tile_np<64>(NP, a, n_tiles, chunk, (hipStream_t)s);
        case 128: return dispatch_tile_np<128>(NP, a, n_tiles, chunk, (hipStream_t)s);
    }
    return RGCN_ERR_WIDTH;
}

}  // namespace rgcn

#ifdef RGCN_STAMPS
extern "C" int rgcn_debug_set_stamps(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(rgcn::g_stamps), &p, sizeof(p));
}
#endif
