import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'amt-saga_amd')]
import numpy as np, torch
from amt_saga import rdcnn
from oracle import rdcnn as orc
for shape, k in (((36, 8), (2, 2)), ((36, 8), (4, 2)), ((40, 8), (2, 2)), ((20, 30), (2, 2)), ((36, 8), (2, 4))):
    try:
        net = rdcnn.res_net(input_shapes=[shape + (1,)], output_classes=1, output_range=[0, 10], kernel_sizes=[k],
                            pool_sizes=[(2, 2)], convolutional_layer_count=2, feature_expand_frequency=0,
                            pool_layer_frequency=0, residual_layer_frequencies=0, weight_seed=5)
    except Exception as e:
        print(shape, k, 'ctor', e); continue
    x = (np.random.default_rng(1).random((3,) + shape) ** 2).astype(np.float32)
    ref = orc.forward(net.weights, net.cfg, [x[..., None]], np.float32, return_logits=True)
    for mode in (0, 1):
        net.set_mode(mode)
        try:
            y, lg = net.predict_device([torch.from_numpy(x).cuda()], return_logits=True)
            lg = lg.cpu().numpy()
            print(shape, k, 'mode', mode, 'nan', int(np.isnan(lg).sum()), 'err', float(np.nanmax(np.abs(lg - ref))))
        except Exception as e:
            print(shape, k, 'mode', mode, 'EXC', e)
