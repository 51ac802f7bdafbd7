// Profiling build of one conv configuration with per-workgroup phase timestamps.
// hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -DYAMD_CONV_TRACE -I../yolo_amd/csrc -I../include conv_trace.hip ../yolo_amd/csrc/yamd_core.hip -o libconv_trace.so
#include "yamd_conv_kernel.h"
using namespace yamd;
namespace yamd { int nms_init_current_device(int) { return 0; } int stem_init_current_device(int) { return 0; } }

#ifndef TR_KS
#define TR_KS 1
#define TR_S 1
#define TR_TW 20
#define TR_MREP 2
#define TR_NREP 4
#define TR_CK 64
#endif
#ifndef TR_NW
#define TR_NW 4
#endif
#ifndef TR_STAGES
#define TR_STAGES 1
#endif

extern "C" int conv_trace_run(const void *x, int x_cs, const void *w, const float *bias, void *y, int y_cs, int batch,
                              int h, int wd, int cin, int cout, unsigned long long *trace, int *grid_out)
{
    constexpr int KS = TR_KS, S = TR_S, TW = TR_TW, MREP = TR_MREP, NREP = TR_NREP, CK = TR_CK;
    ConvParams p;
    p.x = (const __bf16 *)x; p.w = (const char *)w; p.bias = bias; p.res = nullptr; p.y = y; p.y2 = nullptr; p.y2_cs = 0; p.split = cout;
    p.x_cs = x_cs; p.res_cs = 0; p.y_cs = y_cs; p.H = h; p.W = wd;
    const int pad = KS / 2;
    p.Hout = (h + 2 * pad - KS) / S + 1; p.Wout = (wd + 2 * pad - KS) / S + 1; p.Cout = cout; p.act = YAMD_ACT_SILU; p.vin = p.vout = p.vbatch = 0; p.res_scale = 1.0f;
    const int th_max = (TR_NW * 32 * MREP) / TW;
    p.tiles_x = ceil_div(p.Wout, TW); p.tiles_y = ceil_div(p.Hout, th_max); p.TH = ceil_div(p.Hout, p.tiles_y);
    p.n_tiles = ((cout + NREP * 32 - 1) / (NREP * 32)); p.nchunks = cin / CK;
    p.x2 = nullptr; p.x2_cs = 0; p.nsplit = p.nchunks; p.half1 = p.half2 = 0; p.oH = h; p.oW = wd;
#ifdef YAMD_CONV_TRACE
    p.trace = trace;
#endif
    *grid_out = (int)(((long long)batch * p.tiles_y * p.tiles_x * p.n_tiles + 7) / 8 * 8);
    return launch_conv<KS, S, TW, MREP, NREP, CK, false, YAMD_ACT_SILU, TR_NW, TR_STAGES>(p, batch, nullptr);
}
