// A recorded layer program replayed by one call: the whole conv graph of a model (stem, fused convs, pools, upsampling,
// channel copies) as a flat list of the library's own entry points with every argument resolved when the plan is made.
// A caller in any language creates the plan once per (model, batch size, device) and then pays one foreign call per
// batch instead of one per layer; the per-layer argument checks still run (inside the entry points the plan calls).
// Host code only: the plan owns no device memory and launches nothing itself.
#include "yamd_common.h"

#include <new>
#include <vector>

using namespace yamd;

struct yamd_plan {
    std::vector<yamd_plan_op> ops;
    int batch, img_h, img_w;
};

namespace {

int run_op(const yamd_plan &pl, const yamd_plan_op &o, const float *images, void *st)
{
    const int32_t *i = o.i;
    void *const *p = o.p;
    switch (o.kind) {
    case YAMD_OP_CONV:
        return yamd_conv2d_bias_act(&o.desc, p[0], i[0], p[1], (const float *)p[2], p[3], i[1], p[4], i[2], pl.batch, i[3], i[4], st);
    case YAMD_OP_CONV_SPLIT:
        return yamd_conv2d_bias_act_split(&o.desc, p[0], i[0], p[1], (const float *)p[2], p[4], i[2], p[5], i[5], i[6], pl.batch,
                                          i[3], i[4], st);
    case YAMD_OP_CONV_CAT2:
        return yamd_conv2d_bias_act_cat2(&o.desc, p[0], i[0], i[7], i[8] & 1, p[3], i[1], (i[8] >> 1) & 1, p[1], (const float *)p[2],
                                         p[4], i[2], p[5], i[5], i[6], pl.batch, i[3], i[4], st);
    case YAMD_OP_BOTTLENECK:
        return yamd_bottleneck(i[1], i[5], i[6], i[7], p[0], i[0], p[1], (const float *)p[2], i[8], p[4], i[2], pl.batch, i[3], i[4], st);
    case YAMD_OP_C3:
        return yamd_c3_fused(i[1], i[5], i[6], i[7], p[0], i[0], p[1], (const float *)p[2], i[8], p[4], i[2], pl.batch, i[3], i[4], st);
    case YAMD_OP_STEM:
        return yamd_stem_conv(i[0], images, pl.batch, pl.img_h, pl.img_w, p[1], (const float *)p[2], i[1], i[2], p[4], i[3], st);
    case YAMD_OP_STEM_CONV2:
        return yamd_stem_conv2(i[0], images, pl.batch, pl.img_h, pl.img_w, p[1], (const float *)p[2], i[1], i[2], p[5],
                               (const float *)p[6], i[4], i[5], p[4], i[3], st);
    case YAMD_OP_FOCUS:
        return yamd_focus_pack(images, p[4], i[0], pl.batch, pl.img_h, pl.img_w, st);
    case YAMD_OP_NHWC16:
        return yamd_nchw_to_nhwc16(images, p[4], i[0], pl.batch, i[1], pl.img_h, pl.img_w, st);
    case YAMD_OP_UPSAMPLE2X:
        return yamd_upsample2x(p[0], i[0], p[4], i[1], pl.batch, i[2], i[3], i[4], st);
    case YAMD_OP_MAXPOOL5_CASCADE:
        return yamd_maxpool5_cascade(p[0], i[0], p[4], p[5], p[6], i[1], pl.batch, i[2], i[3], i[4], st);
    case YAMD_OP_MAXPOOL3_CASCADE:
        return yamd_maxpool3_cascade(p[0], i[0], p[4], p[5], p[6], i[1], pl.batch, i[2], i[3], i[4], st);
    case YAMD_OP_MAXPOOL2:
        return yamd_maxpool2(p[0], i[0], p[4], i[1], pl.batch, i[2], i[3], i[4], st);
    case YAMD_OP_DEPTH_TO_SPACE2:
        return yamd_depth_to_space2(p[0], i[0], p[4], i[1], pl.batch, i[2], i[3], i[4], st);
    case YAMD_OP_COPY_CHANNELS:
        return yamd_copy_channels(p[0], i[0], p[4], i[1], (long long)pl.batch * i[2] * i[3], i[4], st);
    default:
        break;
    }
    YAMD_REQUIRE(false, "unknown plan op kind %d", o.kind);
    return YAMD_EINVAL;
}

}  // namespace

extern "C" int yamd_plan_create(const yamd_plan_op *ops, int n_ops, int batch, int img_h, int img_w, yamd_plan **plan_out)
{
    YAMD_REQUIRE(ops && plan_out && n_ops > 0 && n_ops <= 65536, "bad argument (n_ops=%d)", n_ops);
    YAMD_REQUIRE(batch > 0 && img_h > 0 && img_w > 0, "bad shape %dx%dx%d", batch, img_h, img_w);
    for (int k = 0; k < n_ops; ++k)
        YAMD_REQUIRE(ops[k].kind >= YAMD_OP_CONV && ops[k].kind <= YAMD_OP_C3, "op %d: unknown kind %d", k, ops[k].kind);
    yamd_plan *pl = new (std::nothrow) yamd_plan;
    YAMD_REQUIRE(pl != nullptr, "out of host memory");
    pl->ops.assign(ops, ops + n_ops);
    pl->batch = batch;
    pl->img_h = img_h;
    pl->img_w = img_w;
    *plan_out = pl;
    return YAMD_OK;
}

extern "C" int yamd_plan_run(const yamd_plan *plan, const float *images, void *stream, int *failed_op)
{
    if (failed_op) *failed_op = -1;
    YAMD_REQUIRE(plan && images, "null argument");
    for (size_t k = 0; k < plan->ops.size(); ++k) {
        const int rc = run_op(*plan, plan->ops[k], images, stream);
        if (rc != YAMD_OK) {
            if (failed_op) *failed_op = (int)k;
            return rc;      // the failing entry point has set the message
        }
    }
    return YAMD_OK;
}

extern "C" int yamd_plan_size(const yamd_plan *plan)
{
    return plan ? (int)plan->ops.size() : 0;
}

extern "C" void yamd_plan_destroy(yamd_plan *plan)
{
    delete plan;
}
