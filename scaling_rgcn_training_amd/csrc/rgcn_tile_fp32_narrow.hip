// rgcn_tile_fp32_narrow.hip -- the instantiations of rgcn_tile_kernel for gathered widths 16 and 32 (rgcn_tile_fp32_kernel.h).
#include "rgcn_tile_fp32_kernel.h"

namespace rgcn {

int dispatch_tile_narrow(int KP, int NP, const TileArgs& a, int n_tiles, int chunk, void* s) {
    switch (KP) {
        case 16: return dispatch_tile_np<16>(NP, a, n_tiles, chunk, (hipStream_t)s);
        case 32: return dispatch_tile_np<32>(NP, a, n_tiles, chunk, (hipStream_t)s);
    }
    return RGCN_ERR_WIDTH;
}

}  // namespace rgcn

#ifdef RGCN_STAMPS
extern "C" int rgcn_debug_set_stamps_narrow(unsigned long long* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(rgcn::g_stamps), &p, sizeof(p));
}
#endif
