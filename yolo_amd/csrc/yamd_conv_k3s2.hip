// Instantiations of the implicit-GEMM conv kernel for one (kernel size, stride) family.
#include "yamd_conv_kernel.h"

namespace yamd {
int conv_dispatch_k3s2(const ConvParams &p, int batch, int tw, int mrep, int nrep, int ck, int f32, int nwaves, hipStream_t st)
{
    if (mrep == 1 && nrep == 1 && ck == 16 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<3, 2, 1, 1, 16, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 2 && ck == 16 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<3, 2, 1, 2, 16, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 4 && ck == 16 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<3, 2, 1, 4, 16, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 2 && ck == 32 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<3, 2, 1, 2, 32, false, 4>(p, batch, tw, st);
    if (mrep == 1 && nrep == 2 && ck == 16 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<3, 2, 1, 2, 16, false, 8>(p, batch, tw, st);
    if (mrep == 1 && nrep == 4 && ck == 16 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<3, 2, 1, 4, 16, false, 8>(p, batch, tw, st);
    if (mrep == 1 && nrep == 2 && ck == 32 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<3, 2, 1, 2, 32, false, 8>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 16 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<3, 2, 2, 4, 16, false, 8>(p, batch, tw, st);
    if (mrep == 2 && nrep == 2 && ck == 16 && f32 == 0 && nwaves == 8)
        return launch_conv_variant<3, 2, 2, 2, 16, false, 8>(p, batch, tw, st);
    if (mrep == 2 && nrep == 4 && ck == 16 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<3, 2, 2, 4, 16, false, 4>(p, batch, tw, st);
    if (mrep == 2 && nrep == 2 && ck == 16 && f32 == 0 && nwaves == 4)
        return launch_conv_variant<3, 2, 2, 2, 16, false, 4>(p, batch, tw, st);
    return 1;   // not built
}
}  // namespace yamd
