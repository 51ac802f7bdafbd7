import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import postprocess as pp
from yolo_amd import torch_utils as tu
G = np.load('tests/golden/nms.npz')
pred = G['single_xyxy__pred'][:1]
img, cx, conf, nms, P, pc = G['single_xyxy__meta'].tolist()
want, widx = pp.non_max_suppression(pred, img, 'xyxy', conf, nms, return_index=True)
for rep in range(3):
    rows, counts, index = tu.nms_batched(torch.from_numpy(pred).cuda(), img, 'xyxy', conf, nms, 0, want_index=True)
    k = int(counts[0]); gi = index[0, :k].cpu().numpy().tolist()
    print(rep, k, len(widx[0]))
    w = widx[0].tolist()
    print(' missing', [x for x in w if x not in gi])
    print(' extra', [x for x in gi if x not in w])
# rank of missing in sorted order
p = pred[0]; sc = p[:, 4] * p[:, 5]
cand = np.where(sc >= np.float32(conf))[0]
order = cand[np.argsort(-sc[cand], kind='stable')].tolist()
print('ranks of missing', [order.index(x) for x in w if x not in gi])
print('ranks of kept', [order.index(x) for x in gi])
