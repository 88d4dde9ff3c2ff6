"""GPU parity of the training step (amt_trainer_step: training-mode BN, backward, Adagrad; res_net.train /
.test, RDCNN.py:503-589) against the numpy training oracle (oracle/train.py, itself pinned to PyTorch
autograd in tests/test_oracle_train.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_oracle_train import CASES      # noqa: E402  (the shallow topologies: every layer kind)


@pytest.fixture(scope='module')
def env():
    import torch
    assert torch.cuda.is_available()
    from amt_saga import rdcnn, heads, hyperparams
    from oracle import train as otr, rdcnn as orc
    return dict(torch=torch, rdcnn=rdcnn, heads=heads, hp=hyperparams, otr=otr, orc=orc)


def _batch(cfg, B, seed):
    rng = np.random.default_rng(seed)
    xs = [(rng.random((B,) + tuple(s[:2]) + (1,)) ** 2).astype(np.float32) for s in cfg['input_shapes']]
    if cfg['output_classes'] > 1:
        y = rng.integers(0, cfg['output_classes'], B).astype(np.float64)
    else:
        lo, hi = cfg['output_range']
        y = rng.uniform(lo, hi, B)
    return xs, y


def _opt(net):
    """(lr, epsilon, initial accumulator) the product will train with: the Keras-generation triple, overridden by
    the individual attributes (rdcnn.res_net.ADAGRAD_DEFAULTS)."""
    lr0, eps0, acc0 = net.ADAGRAD_DEFAULTS[net.keras_optimizer_version]
    return (getattr(net, 'learning_rate', lr0), getattr(net, 'adagrad_epsilon', eps0),
            getattr(net, 'adagrad_initial_accumulator', acc0))


def _to_act(net, y):
    return net._scale_output_to_activation(y) if net.output_classes == 1 else y


def _compare_step(env, net, xs, y, w_before, acc, tag):
    """One train() on the device vs one oracle train_on_batch from the same state; returns the oracle's state."""
    pred = net.train(xs if len(xs) > 1 else xs[0], y)
    loss = net.metrics_train[-1][0]
    grads = net.gradients()
    o_loss, o_pred, g, stats = env['otr'].forward_backward(w_before, net.cfg, xs, _to_act(net, y), np.float32)
    lr, eps, acc0 = _opt(net)
    o_loss2, _, w_after, acc2 = env['otr'].train_on_batch(w_before, net.cfg, xs, _to_act(net, y), acc=acc, lr=lr, eps=eps,
                                                          dtype=np.float32, initial_accumulator=acc0)
    assert abs(loss - o_loss) <= 2e-5 * max(abs(o_loss), 1e-3), (tag, loss, o_loss)
    o_out = net._scale_activation_to_output(o_pred) if net.output_classes == 1 else o_pred
    assert np.abs(pred - o_out).max() <= 1e-4 * max(np.abs(o_out).max(), 1e-6), tag
    # gradients: against the float64 oracle, to 3e-4 of the tensor's largest gradient plus four times the distance
    # the float32 oracle itself keeps from float64 (the arithmetic noise floor: a conv bias in front of a
    # training-mode BN has a structurally zero gradient, and both float32 results are pure rounding noise there)
    _, _, g64, _ = env['otr'].forward_backward(w_before, net.cfg, xs, _to_act(net, y), np.float64)
    gmax = max(np.abs(v).max() for v in g64.values())
    for k, gk in g64.items():
        err = np.abs(grads[k] - gk).max()
        noise = np.abs(g[k] - gk).max()
        assert err <= 3e-4 * np.abs(gk).max() + 4 * noise + 1e-7 * gmax, (tag, k, err, np.abs(gk).max(), noise, gmax)
    net._sync_from_trainer()
    for k, gk in g.items():
        if np.abs(gk).max() < 1e-4 * gmax:
            continue                      # structurally zero gradient (a bias in front of a training-mode BN): Adagrad
        #                                   divides noise by noise there, in Keras as here
        big = np.abs(gk) > 1e-3 * np.abs(gk).max()          # elements whose sign and size are not rounding noise
        assert np.abs(net.weights[k] - w_after[k])[big].max() <= 0.02 * lr, (tag, k)
    for p_ in stats:
        for leaf in ('mean', 'var'):
            a, b = net.weights[p_ + '/' + leaf], w_after[p_ + '/' + leaf]
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-3), (tag, p_, leaf)
    return w_after, acc2


@pytest.mark.parametrize('case', range(len(CASES)))
def test_train_step_vs_oracle(env, case):
    net = env['rdcnn'].res_net(weight_seed=31 + case, calibrated=False, **CASES[case])
    # both Keras generations' Adagrad defaults, as consistent triples (lr 0.01 with accumulator 0.1 is no
    # version's default)
    net.keras_optimizer_version = 'keras-2.2' if case % 2 == 0 else 'tf.keras-1.14'
    assert _opt(net) == ((0.01, 1e-7, 0.0) if case % 2 == 0 else (0.001, 1e-7, 0.1))
    xs, y = _batch(net.cfg, 6, case)
    w0 = {k: v.copy() for k, v in net.weights.items()}
    w1, acc = _compare_step(env, net, xs, y, w0, None, 'step1')
    # second step from the device's own state (accumulators carry over); the oracle restarts from the device's weights
    # so that rounding-level differences of step 1 do not compound into the comparison
    dev_w1 = {k: v.copy() for k, v in net.weights.items()}
    xs2, y2 = _batch(net.cfg, 6, case + 100)
    _compare_step(env, net, xs2, y2, dev_w1, acc, 'step2')
    # test(): inference-mode forward (moving statistics) + loss, no update
    net._sync_from_trainer()
    w_now = {k: v.copy() for k, v in net.weights.items()}
    pred = net.test(xs2 if len(xs2) > 1 else xs2[0], y2)
    ref = env['orc'].forward(w_now, net.cfg, xs2, np.float32)
    assert np.abs(pred - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6)
    net._sync_from_trainer()
    assert all(np.array_equal(net.weights[k], w_now[k]) for k in w_now)          # nothing moved
    yp = net.predict(xs2 if len(xs2) > 1 else xs2[0])
    assert np.abs(yp - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6)


def test_train_batch_size_changes_between_steps(env, monkeypatch):
    """The trainer (re)allocates its activation / gradient / reduction buffers when a larger batch arrives and keeps
    weights, Adagrad accumulators and moving statistics across that: steps of 3, 7 and 2 windows on one net (fast forms
    on, so the per-window maxima and the partial-sum buffers are resized too), each against the oracle from the
    device's own state."""
    monkeypatch.setenv('AMT_TRAIN_FAST_MIN_M', '0')
    net = env['rdcnn'].res_net(weight_seed=77, calibrated=False, **CASES[1])
    acc = None
    for i, B in enumerate((3, 7, 2)):
        xs, y = _batch(net.cfg, B, 900 + i)
        net._sync_from_trainer() if i else None
        w_dev = {k: v.copy() for k, v in net.weights.items()}
        _, acc = _compare_step(env, net, xs, y, w_dev, acc, 'batch %d' % B)


def test_velocity_head_first_step_and_learning(env):
    """The reference's smallest head (11 conv layers, 36 x 8 input) on a batch of 8 (main.py -batch_size): the first
    train_on_batch matches the oracle; and forty steps on a fixed batch drive the loss of two shallow nets (one
    regression / MSE, one softmax / sparse-CCE) down by an order of magnitude, tracking the oracle's trajectory."""
    p = env['hp'].Hyperparams(N=2048)
    h = env['heads'].VelocityClassifier(p)
    rng = np.random.default_rng(4)
    specs = [(rng.random((36, 8)) ** 2).astype(np.float32) for _ in range(8)]
    gold = list(rng.uniform(5, 125, 8))
    x = np.stack(specs)[..., None]
    w0 = {k: v.copy() for k, v in h.weights.items()}
    _compare_step(env, h, [x], np.array(gold), w0, None, 'velocity')
    pred = h.classify(specs, gold)                            # classify(spec, gold) trains (velocity_classifier.py:48-55)
    assert pred.shape == (8, 1) and len(h.metrics_train) == 2 and h.metrics_train[1][0] < h.metrics_train[0][0]
    for case in (0, 1):
        net = env['rdcnn'].res_net(weight_seed=31 + case, calibrated=False, **CASES[case])
        # an explicit optimiser setting for the learning demonstration (the attributes override the Keras-generation
        # triple): with a zero initial accumulator the first Adagrad steps are +-lr for EVERY weight whatever its
        # gradient, and forty of them do not bring a fresh net's loss down
        net.learning_rate, net.adagrad_initial_accumulator = 0.01, 0.1
        xs, y = _batch(net.cfg, 8, 7)
        w, acc, ref = net.weights, None, []
        for _ in range(40):
            net.train(xs[0], y)
        for _ in range(40):
            lr, eps, acc0 = _opt(net)
            loss, _, w, acc = env['otr'].train_on_batch(w, net.cfg, xs, _to_act(net, y), acc=acc, lr=lr, eps=eps,
                                                        dtype=np.float32, initial_accumulator=acc0)
            ref.append(loss)
        losses = [m[0] for m in net.metrics_train]
        assert losses[-1] < 0.15 * losses[0], (case, losses[0], losses[-1])
        # same trajectory as the oracle (rounding differences grow slowly over the steps)
        assert np.abs(np.array(losses) - np.array(ref)).max() <= 0.02 * losses[0], (case, losses[-3:], ref[-3:])


def test_timing_head_train_on_batch_vs_oracle(env):
    """The head that is 97 % of the flops, through one train_on_batch (RDCNN.py:503-526 via timing_classifier.py:13-36)
    at the reference's default N = 4096 (20 x 258 input, 33 layers, 4 x 16 kernels, 2 x 8 pools, projected shortcuts at
    layers 14 and 26), batch of 2: loss, predictions, every gradient, the Adagrad update and the moving statistics
    against oracle/train.py -- the same bars as the shallow nets and the velocity head."""
    p = env['hp'].Hyperparams(N=4096)
    h = env['heads'].timming_classifier(p, calibrated=False)
    assert h.cfg['input_shapes'][0] == (20, 258, 1) and h.cfg['convolutional_layer_count'] == 33
    rng = np.random.default_rng(12)
    x = (rng.random((2, 20, 258, 1)) ** 2).astype(np.float32)
    gold = rng.uniform(0, 258, 2)                  # frames (the reference feeds seconds into a frames-ranged head)
    w0 = {k: v.copy() for k, v in h.weights.items()}
    _compare_step(env, h, [x], gold, w0, None, 'timing N=4096 B=2')
    assert len(h.metrics_train) == 1


@pytest.mark.parametrize('case', range(len(CASES)))
def test_train_step_fast_forms_on_small_layers(env, case, monkeypatch):
    """By default layers with fewer than 2048 output positions per batch stay on the GEMM path (a handful of workgroups
    walking the whole contraction is slower there), so the small test graphs above never reach the trainer's forms of
    the split-fp16 kernel for their kernel sizes / the masked small-image form.  AMT_TRAIN_FAST_MIN_M=0 sends every
    covered layer through them: same oracle comparison, same bars."""
    monkeypatch.setenv('AMT_TRAIN_FAST_MIN_M', '0')
    net = env['rdcnn'].res_net(weight_seed=131 + case, calibrated=False, **CASES[case])
    xs, y = _batch(net.cfg, 6, 500 + case)
    w0 = {k: v.copy() for k, v in net.weights.items()}
    _compare_step(env, net, xs, y, w0, None, 'fast-small case %d' % case)


def test_training_paths_agree_at_full_width(env, monkeypatch):
    """The training step's fast forms against its plain ones on the metric shape (timing head, N = 2048: 20 x 516 input,
    batch of 3): split-fp16 forward convolutions / data gradients (conv_f16x3s_kernel, TRAIN form), the weight gradient
    without im2col (wgrad_kernel) and the fused BatchNormalization passes (bnf_*) on one side; im2col + f32-MFMA GEMMs +
    col2im and the generic column reductions on the other (AMT_TRAIN_FAST / AMT_TRAIN_WGRAD / AMT_TRAIN_FUSED_BN = 0,
    read when a trainer is created).  Same weights, same batch: loss, predictions and every gradient must agree to the
    bars the oracle comparison uses (the oracle itself needs minutes at this size)."""
    p = env['hp'].Hyperparams(N=2048)
    rng = np.random.default_rng(5)
    x = (rng.random((3, 20, 516, 1)) ** 2).astype(np.float32)
    gold = rng.uniform(0, 516, 3)
    res = []
    for plain in (False, True):
        for k in ('AMT_TRAIN_FAST', 'AMT_TRAIN_WGRAD', 'AMT_TRAIN_FUSED_BN'):
            if plain:
                monkeypatch.setenv(k, '0')
            else:
                monkeypatch.delenv(k, raising=False)
        h = env['heads'].timming_classifier(p, calibrated=False)
        pred = h.train(x, gold)
        res.append((h.metrics_train[-1][0], pred, h.gradients()))
    (l0, p0, g0), (l1, p1, g1) = res
    assert abs(l0 - l1) <= 2e-5 * max(abs(l1), 1e-3), (l0, l1)
    assert np.abs(p0 - p1).max() <= 1e-4 * max(np.abs(p1).max(), 1e-6)
    gmax = max(np.abs(v).max() for v in g1.values())
    worst = 0.0
    for k in g1:
        err = np.abs(g0[k] - g1[k]).max()
        if np.abs(g1[k]).max() < 1e-4 * gmax:        # structurally zero (a bias in front of a training-mode BN): rounding
            assert np.abs(g0[k]).max() < 1e-4 * gmax, k      # noise on both sides, as in _compare_step
            continue
        bar = 3e-4 * np.abs(g1[k]).max() + 1e-6 * gmax
        worst = max(worst, err / bar)
        assert err <= bar, (k, err, np.abs(g1[k]).max(), gmax)
    print('training paths: worst gradient difference / bar = %.3f' % worst)
