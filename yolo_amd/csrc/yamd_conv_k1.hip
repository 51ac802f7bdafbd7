// Instantiations of the implicit-GEMM conv kernel for one (kernel size, stride) family.
#include "yamd_conv_kernel.h"

namespace yamd {
int conv_dispatch_k1(const ConvParams &p, int batch, int tw, int mrep, int nrep, int ck, int f32, int nwaves, hipStream_t st)
{
    if (mrep == 1 && nrep == 1 && ck == 32 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 1, 1, 32, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 1 && ck == 64 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 1, 1, 64, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 2 && ck == 32 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 1, 2, 32, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 2 && ck == 64 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 1, 2, 64, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 4 && ck == 32 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 1, 4, 32, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 4 && ck == 64 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 1, 4, 64, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 1 && ck == 32 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 1, 32, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 1 && ck == 64 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 1, 64, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 2 && ck == 32 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 2, 32, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 2 && ck == 64 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 2, 64, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 32 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 4, 32, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 64 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 4, 64, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 64 && f32 == 1 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 4, 64, true, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 4 && ck == 64 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<1, 1, 1, 4, 64, false, 8>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 64 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<1, 1, 2, 4, 64, false, 8>(p, batch, tw, st);
    if (mrep == 1 && nrep == 4 && ck == 32 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<1, 1, 1, 4, 32, false, 8>(p, batch, tw, st);
    if (mrep == 1 && nrep == 2 && ck == 64 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<1, 1, 1, 2, 64, false, 8>(p, batch, tw, st);
    if (mrep == 2 && nrep == 2 && ck == 64 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<1, 1, 2, 2, 64, false, 8>(p, batch, tw, st);
    // 16-channel chunks: layers whose input width is a multiple of 16 but not of 32 (the m-size networks: 48, 144 ...)
    if (mrep == 2 && nrep == 1 && ck == 16 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 1, 16, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 16 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 4, 16, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 32 && f32 == 1 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 4, 32, true, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 16 && f32 == 1 && nwaves == 4)
        return launch_conv_variant<1, 1, 2, 4, 16, true, 4>(p, batch, tw, st);
    return 1;   // not built
}
}  // namespace yamd
