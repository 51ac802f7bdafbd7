"""Per-launch parity gate for every model family: each recorded op of the compiled program (conv, fused residual unit,
fused stem, conv over a concatenation, pools, upsampling, depth-to-space) is re-computed in fp32 on the CPU FROM THE
GPU's OWN INPUT of that op (oracle/op_ref.py), so a difference is that launch's bf16 rounding alone and one bad layer
cannot hide inside the end-to-end tolerance.  Bound: relative L2 <= 1e-2 per launch (bf16 operands, fp32 accumulate,
one bf16 rounding of the output: 2e-3 is typical); pools / upsampling / copies exact.  Full benchmark resolution, so the
tilings checked are the ones the benchmark runs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-2


def _rel_l2(got: torch.Tensor, want: torch.Tensor) -> float:
    d = (got.double() - want.double()).norm().item()
    return d / max(want.double().norm().item(), 1e-12)


def gate(name: str, size: int, batch: int, device, seed: int = 11):
    from oracle import op_ref
    from yolo_amd import models
    from yolo_amd.init import seeded_images, seeded_state_dict
    p, m = models.create(name, img_size=size)
    m.load_state_dict(seeded_state_dict(m, seed=seed, mode="spread", **models.init_options(name)))
    imgs = torch.from_numpy(seeded_images(batch, p.img_size, seed=seed))
    prog = m.program()
    prog.run(imgs.to(device))
    torch.cuda.synchronize()
    cache = {}

    def read(v):
        key = (prog.resolve(v), v.c, v.h, v.w)
        if key not in cache:
            cache[key] = prog.read_view(v, batch, device).cpu()
        return cache[key]

    torch.set_num_threads(16)
    worst, n_conv, n_exact = {}, 0, 0
    with torch.no_grad():
        for op in prog.ops:
            for label, (want, got, exact) in op_ref.op_reference(op, read, imgs).items():
                assert want.shape == got.shape, (label, want.shape, got.shape)
                if exact:
                    n_exact += 1
                    assert torch.equal(want, got), f"{name}: {label} is not exact"
                else:
                    n_conv += 1
                    worst[label] = _rel_l2(got, want)
    assert n_conv == prog.n_conv_launches(), (n_conv, prog.n_conv_launches())
    bad = {k: v for k, v in worst.items() if not v <= TOL}
    assert not bad, f"{name}: launches beyond rel-L2 {TOL}: {bad}"
    return worst, n_exact


@pytest.mark.parametrize("name", ["v5s", "v8s", "v3", "v4", "v7pose", "v6face"])
def test_every_launch_of_the_benchmark_models_at_full_size(name, gpu_device):
    worst, _ = gate(name, 0, 2, gpu_device)
    assert len(worst) >= 40 and float(np.median(list(worst.values()))) < 5e-3


@pytest.mark.parametrize("name,size", [("v5face", 128), ("v5pose", 128), ("v7", 128), ("v7x", 128), ("v6", 128), ("v6m", 128),
                                       ("v6l", 128), ("v6posel", 128), ("v5m", 160), ("v8m", 160), ("v8n", 160),
                                       ("v5posex", 128), ("v5facem", 128), ("v6facen", 128)])
def test_every_launch_of_the_other_variants(name, size, gpu_device):
    """Other widths and tasks at a small size (channel counts that are multiples of 16 only, padded stems, DFL heads,
    weighted shortcuts, transposed convs)."""
    gate(name, size, 3, gpu_device)
