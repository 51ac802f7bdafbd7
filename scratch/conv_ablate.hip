// Profiling builds of ONE conv configuration with parts of the chunk loop removed (YAMD_CONV_ABLATE):
//   unset: the product kernel; 1: nothing staged after the first chunk (barriers kept); 2: no MFMAs; 3: no staging and no barriers.
// built by scratch/conv_ablate.py
#include "yamd_conv_kernel.h"
using namespace yamd;
namespace yamd { int nms_init_current_device(int) { return 0; } int stem_init_current_device(int) { return 0; } }

extern "C" int conv_ablate_run(const void *x, int x_cs, const void *w, const float *bias, void *y, int y_cs, int batch,
                               int h, int wd, int cin, int cout)
{
    constexpr int KS = TR_KS, S = TR_S, TW = TR_TW, MREP = TR_MREP, NREP = TR_NREP, CK = TR_CK, NW = TR_NW;
#ifndef TR_STAGES
#define TR_STAGES 1
#endif
    ConvParams p;
    p.x = (const __bf16 *)x; p.w = (const char *)w; p.bias = bias; p.res = nullptr; p.y = y; p.y2 = nullptr; p.y2_cs = 0; p.split = cout;
    p.x_cs = x_cs; p.res_cs = 0; p.y_cs = y_cs; p.H = h; p.W = wd;
    const int pad = KS / 2;
    p.Hout = (h + 2 * pad - KS) / S + 1; p.Wout = (wd + 2 * pad - KS) / S + 1; p.Cout = cout; p.act = YAMD_ACT_SILU;
    p.vin = p.vout = p.vbatch = 0; p.res_scale = 1.0f;
#ifdef YAMD_CONV_SKEW
    p.skew = getenv("CONV_SKEW") ? atoi(getenv("CONV_SKEW")) : 0; p.skew_blocks = getenv("CONV_SKEW_BLOCKS") ? atoi(getenv("CONV_SKEW_BLOCKS")) : 512;
#endif
    const int th_max = (NW * 32 * MREP) / TW;
    p.tiles_x = ceil_div(p.Wout, TW); p.tiles_y = ceil_div(p.Hout, th_max); p.TH = ceil_div(p.Hout, p.tiles_y);
    p.n_tiles = ((cout + NREP * 32 - 1) / (NREP * 32)); p.nchunks = cin / CK;
    p.x2 = nullptr; p.x2_cs = 0; p.nsplit = p.nchunks; p.half1 = p.half2 = 0; p.oH = h; p.oW = wd;
    return launch_conv<KS, S, TW, MREP, NREP, CK, false, YAMD_ACT_SILU, NW, TR_STAGES>(p, batch, nullptr);
}
