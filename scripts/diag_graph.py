"""diagnostic: eager vs HIP-graph replay of the ResNet swap's step, loss per step"""
import sys, torch
sys.path.insert(0, ".")
from oracle import resnet_ref as RR
from tensorflow_yolo2_amd import synthetic
from tensorflow_yolo2_amd.yolo2_nets import tf_resnet
blocks = RR.scaled_blocks(8)
kw = dict(dtype="f32", blocks=blocks, root_depth=8, fc_hidden=512, seed=3)
a = tf_resnet.ResNet50Yolo(2, 64, graph=False, **kw); a.guard = True
b = tf_resnet.ResNet50Yolo(2, 64, graph=True, graph_check_every=100, **kw)
c = tf_resnet.ResNet50Yolo(2, 64, graph=True, graph_check_every=100, **kw); c._eager_on_gstream = -100   # never captures
for i in range(6):
    x = torch.as_tensor(synthetic.images(2, 64, 100 + i)).cuda()
    lab = torch.as_tensor(synthetic.det_labels(2, 64, 2, 200 + i)).cuda()
    la = a.step(x, lab)[0].clone(); lb = b.step(x, lab)[0].clone(); lc = c.step(x, lab)[0].clone()
    torch.cuda.synchronize()
    print(i, [round(v, 4) for v in la.tolist()], [round(v, 4) for v in lb.tolist()], [round(v, 4) for v in lc.tolist()],
          "seed", a.drop_seed, int(b._seed_dev), int(c._seed_dev), "ctrl", a.ctrl[:3].tolist(), b.ctrl[:3].tolist())
