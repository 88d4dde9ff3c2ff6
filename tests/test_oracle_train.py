"""CPU: the training oracle (oracle/train.py: hand-written backward of the RDCNN graph, BN in training mode,
MSE / sparse-CCE, Adagrad) against PyTorch autograd on the same graph in float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import train as otr
from amt_saga.rdcnn import res_net

CASES = [
    dict(input_shapes=[(12, 10, 1)], output_classes=1, output_range=[3, 40], kernel_sizes=[(4, 2)], pool_sizes=[(2, 2)],
         convolutional_layer_count=5, feature_expand_frequency=2, pool_layer_frequency=2, residual_layer_frequencies=2),
    dict(input_shapes=[(9, 70, 1)], output_classes=7, output_range=[0, 1], kernel_sizes=[(4, 16)], pool_sizes=[(2, 8)],
         convolutional_layer_count=4, feature_expand_frequency=2, pool_layer_frequency=2, residual_layer_frequencies=2),
    dict(input_shapes=[(16, 8, 1), (16, 8, 1)], output_classes=5, output_range=[0, 1], kernel_sizes=[(2, 2), (2, 2)],
         pool_sizes=[(2, 2), (2, 2)], convolutional_layer_count=4, feature_expand_frequency=2, pool_layer_frequency=2,
         residual_layer_frequencies=2),
    dict(input_shapes=[(11, 9, 1)], output_classes=1, output_range=[0, 9], kernel_sizes=[(4, 2)], pool_sizes=[(2, 2)],
         convolutional_layer_count=3, feature_expand_frequency=0, pool_layer_frequency=0, residual_layer_frequencies=0),
]


def _torch_loss_and_grads(w, cfg, xs, y):
    tw = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=k.rsplit('/', 1)[1] in ('kernel', 'bias', 'gamma', 'beta'))
          for k, v in w.items()}

    def bn(x, p):                                         # training mode: batch statistics, biased variance
        return F.batch_norm(x, None, None, tw[p + '/gamma'], tw[p + '/beta'], True, 0.0, 1e-3)

    def conv(x, p):
        k = tw[p + '/kernel']
        kh, kw = k.shape[:2]
        pt, pl = (kh - 1) // 2, (kw - 1) // 2
        return F.conv2d(F.pad(x, (pl, kw - 1 - pl, pt, kh - 1 - pt)), k.permute(3, 2, 0, 1), tw[p + '/bias'])

    r = cfg['residual_layer_frequencies'][0] if cfg['residual_layer_frequencies'] else 0
    flats = []
    for t, x in enumerate(xs):
        p1 = torch.tensor(np.asarray(x, np.float64)).permute(0, 3, 1, 2)
        p0 = p1
        for i in range(1, cfg['convolutional_layer_count'] + 1):
            p1 = torch.sigmoid(bn(conv(p1, 't%d/conv%d' % (t, i)), 't%d/bn%d' % (t, i)))
            if r and i % r == 0:
                a = p0
                if a.shape != p1.shape:
                    if a.shape[1] != p1.shape[1]:
                        a = conv(a, 't%d/sc%d' % (t, i))
                    if a.shape[2:] != p1.shape[2:]:
                        a = F.avg_pool2d(a, (a.shape[2] // p1.shape[2], a.shape[3] // p1.shape[3]))
                    a = bn(a, 't%d/scbn%d' % (t, i))
                p1 = bn(a + p1, 't%d/resbn%d' % (t, i))
                p0 = p1
            if cfg['pool_layer_frequency'] and i % cfg['pool_layer_frequency'] == 0:
                p1 = F.max_pool2d(p1, tuple(cfg['pool_sizes'][t]))
        flats.append(p1.permute(0, 2, 3, 1).reshape(p1.shape[0], -1))
    h = torch.sigmoid(torch.cat(flats, 1) @ tw['dense1/kernel'] + tw['dense1/bias'])
    z2 = h @ tw['dense2/kernel'] + tw['dense2/bias']
    if cfg['output_classes'] > 1:
        loss = F.cross_entropy(z2, torch.tensor(np.asarray(y, np.int64)))
    else:
        loss = ((torch.sigmoid(z2)[:, 0] - torch.tensor(np.asarray(y, np.float64))) ** 2).mean()
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in tw.items() if v.requires_grad}


@pytest.mark.parametrize('case', range(len(CASES)))
def test_backward_matches_autograd(case):
    kw = CASES[case]
    net = res_net(weight_seed=31 + case, calibrated=False, **kw)
    cfg = net.cfg
    rng = np.random.default_rng(case)
    B = 4
    xs = [rng.random((B,) + tuple(s[:2]) + (1,)) ** 2 for s in cfg['input_shapes']]
    y = rng.integers(0, cfg['output_classes'], B) if cfg['output_classes'] > 1 else rng.random(B)
    loss, pred, g, stats = otr.forward_backward(net.weights, cfg, xs, y, np.float64)
    tl, tg = _torch_loss_and_grads(net.weights, cfg, xs, y)
    assert abs(loss - tl) < 1e-12 * max(abs(tl), 1.0)
    assert set(g) == set(tg)
    gmax = max(np.abs(v).max() for v in tg.values())
    for k in tg:
        # (a conv bias in front of a training-mode BN has a structurally zero gradient: floor relative to the largest)
        scale = np.abs(tg[k]).max()
        assert np.abs(g[k] - tg[k]).max() < 1e-8 * scale + 1e-10 * gmax, (k, np.abs(g[k] - tg[k]).max(), scale)
    # every BN layer reported its batch statistics; one Adagrad step moves every trainable tensor by ~lr
    loss32, pred32, w2, acc = otr.train_on_batch(net.weights, cfg, xs, y, dtype=np.float64, initial_accumulator=0.0)
    for k in tg:
        step = np.abs(w2[k] - net.weights[k])
        assert step.max() <= 0.01 + 1e-6
        if np.abs(tg[k]).max() > 1e-6 * gmax:
            assert (step > 0.009).mean() > 0.5, k          # first Adagrad step: lr * g / (|g| + eps) ~ lr * sign(g)
    for p, (mu, var) in stats.items():
        assert np.allclose(w2[p + '/mean'], 0.99 * net.weights[p + '/mean'] + 0.01 * mu, atol=1e-6)
    # a second step with the accumulators shrinks the update (Adagrad)
    _, _, w3, _ = otr.train_on_batch(w2, cfg, xs, y, acc=acc, dtype=np.float64, initial_accumulator=0.0)
    k = 'dense1/kernel'
    assert np.abs(w3[k] - w2[k]).mean() < np.abs(w2[k] - net.weights[k]).mean()
