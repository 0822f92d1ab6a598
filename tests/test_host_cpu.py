"""Host-side mirror of the reference surface that needs no GPU: trajectory export, the lazy reference view's bookkeeping."""
import os
import types

import numpy as np
import torch

import trajopt_grpo_amd as tg
from conftest import GOLDEN, load_golden


def test_save_trajectory_writes_the_reference_csv_byte_for_byte(tmp_path):
    """Rollout_Buffer.save_trajectory (buffers/rollout_buffer.py:72-102) on the golden CartPole rollout against the CSV the
    reference itself wrote for the same tensors (tests/golden/trajectory_cartpole_reset.csv, oracle/tools/gen_goldens.py)."""
    g = load_golden("rollout_cartpole.npz")
    buf = tg.Rollout_Buffer(types.SimpleNamespace(env_fn=lambda: None))
    buf.store(*(torch.from_numpy(g[f"reset_{k}"]) for k in ("obs", "act", "rew", "len", "mask")))
    buf.limit_reference_view(max_episodes=1)                   # must not affect the export
    buf.save_trajectory(str(tmp_path))
    got = open(os.path.join(tmp_path, "trajectory.csv")).read()
    want = open(os.path.join(GOLDEN, "trajectory_cartpole_reset.csv")).read()
    assert got == want
    lines = got.splitlines()
    assert lines[0] == "episode_id," + ",".join(f"observation_{i}" for i in range(5)) + ",action_0"
    assert len(lines) - 1 == int(g["reset_len"].sum())
    assert float(buf.avg_reward[-1]) == float(torch.from_numpy(g["reset_rew"]).sum(2).mean())


def test_retrieve_returns_the_five_stored_tensors():
    g = load_golden("rollout_cartpole.npz")
    buf = tg.Rollout_Buffer(types.SimpleNamespace(env_fn=lambda: None))
    t = tuple(torch.from_numpy(g[f"reset_{k}"]) for k in ("obs", "act", "rew", "len", "mask"))
    buf.store(*t)
    assert all(a is b for a, b in zip(buf.retrieve(), t)) and buf.group_observations is t[0]


def _emulate_f32_chain(fs, x):
    """What tg_mlp_f32_forward computes from mlp.F32ChainStream's stream, restated on the host from the kernel's documented
    operand layout (v_mfma_f32_32x32x2_f32: A lane (i, kk) = A[i][kk], B lane (j, kk) = B[kk][j]; accumulator register r of lane
    half h of tile mt is feature 32 mt + F(r, h)): fp64, one row."""
    H, nh, K2, MT = fs.H, fs.n_hidden, fs.in_pad // 2, fs.H // 32
    st = fs.stream.double().numpy()
    o = 0
    w0 = st[o:o + H * fs.in_pad].reshape(MT, K2 // 4, 64, 4); o += H * fs.in_pad
    bias = st[o:o + nh * H].reshape(nh, H); o += nh * H
    wh = st[o:o + 4 * H].reshape(4, H); o += 4 * H
    bh = st[o:o + 4]; o += 4
    n_hh = nh - 1
    blocks = st[o:].reshape(2 * n_hh, MT, H // 8, 64, 4)
    F = lambda t, kk: (t & 3) + 8 * (t >> 2) + 4 * kk
    xp = np.zeros(fs.in_pad)
    xp[:fs.in_dim] = x
    a = np.zeros(H)
    for mo in range(MT):
        for i in range(32):
            a[32 * mo + i] = bias[0, 32 * mo + i] + sum(w0[mo, s // 4, i + 32 * kk, s % 4] * xp[kk * K2 + s] for s in range(K2) for kk in range(2))
    a = np.maximum(a, 0)
    acts = [a]
    for l in range(1, nh):
        nxt = np.zeros(H)
        for mo in range(MT):
            blk = blocks[(l - 1), mo]
            for i in range(32):
                nxt[32 * mo + i] = bias[l, 32 * mo + i] + sum(blk[s // 4, i + 32 * kk, s % 4] * a[32 * (s // 16) + F(s % 16, kk)]
                                                                for s in range(H // 2) for kk in range(2))
        a = np.maximum(nxt, 0)
        acts.append(a)
    out = wh @ a + bh
    # backward-data blocks: dA_{l-1} = W_l^T dZ_l, top layer first
    dz = np.arange(H, dtype=np.float64) / H - 0.3
    back = []
    for k, l in enumerate(range(nh - 1, 0, -1)):
        nxt = np.zeros(H)
        for ko in range(MT):
            blk = blocks[n_hh + k, ko]
            for i in range(32):
                nxt[32 * ko + i] = sum(blk[s // 4, i + 32 * kk, s % 4] * dz[32 * (s // 16) + F(s % 16, kk)] for s in range(H // 2) for kk in range(2))
        back.append((l, dz.copy(), nxt))
        dz = nxt
    return out, acts, back


def test_f32_chain_weight_stream_layout():
    """mlp.F32ChainStream (the operand order tg_mlp_f32_forward / _forward_backward consume) against torch on the CPU: forward
    blocks, the first layer's padded k pairs, the natural-order tables and the transposed (backward) blocks."""
    from trajopt_grpo_amd import mlp as M
    for S, A, hidden in [(5, 1, (64, 64)), (20, 4, (128, 128, 128)), (9, 2, (64,))]:
        torch.manual_seed(S)
        net = tg.NeuralNetwork(S, A, hidden, "ReLU")
        H = M.f32_chain_supported(net)
        assert H == hidden[0]
        fs = M.F32ChainStream(net, H)
        x = torch.randn(S)
        out, acts, back = _emulate_f32_chain(fs, x.numpy().astype(np.float64))
        lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        h = x.double()
        for l, a in zip(lin[:-1], acts):
            h = torch.relu(l.weight.double() @ h + l.bias.double())
            np.testing.assert_allclose(a, h.detach().numpy(), rtol=1e-12, atol=1e-12)
        ref = (lin[-1].weight.double() @ h + lin[-1].bias.double()).detach().numpy()
        np.testing.assert_allclose(out[:A], ref, rtol=1e-12, atol=1e-12)
        assert np.all(out[A:] == 0)
        for l, dz, got in back:
            np.testing.assert_allclose(got, (lin[l].weight.double().t() @ torch.from_numpy(dz)).detach().numpy(), rtol=1e-12, atol=1e-12)
    assert M.f32_chain_supported(tg.NeuralNetwork(5, 1, (32, 32), "ReLU")) == 0
    assert M.f32_chain_supported(tg.NeuralNetwork(5, 8, (64, 64), "ReLU")) == 0
    assert M.f32_chain_supported(tg.NeuralNetwork(5, 1, (128,) * 5, "ReLU")) == 0


def test_f32_resident_weight_stream_layout():
    """mlp.F32ResStream (the operands of tg_mlp_f32r_forward[_backward], v_mfma_f32_16x16x4_f32: A lane (i, g) = A[i][g], B lane
    (j, g) = B[g][j], accumulator register r of lane (j, g) of tile t = feature 16 t + 4 g + r of row j) against torch on the CPU."""
    from trajopt_grpo_amd import mlp as M
    for S, A, hidden in [(5, 1, (128, 128)), (20, 4, (128, 128)), (9, 2, (128,))]:
        torch.manual_seed(S)
        net = tg.NeuralNetwork(S, A, hidden, "ReLU")
        assert M.f32_res_supported(net) == 128
        fs = M.F32ResStream(net)
        H, NT, K4, nh = 128, 8, fs.in_pad // 4, len(hidden)
        st = fs.stream.double().numpy()
        n_stream = 2 * (nh - 1) * NT * NT * 256
        blocks = st[:n_stream].reshape(2 * (nh - 1), NT, NT, 64, 4)
        w0 = fs.w0.double().numpy().reshape(NT, K4, 64)
        tab = fs.table.double().numpy()
        bias, wh, bh = tab[:2 * H].reshape(2, H), tab[2 * H:6 * H].reshape(4, H), tab[6 * H:6 * H + 4]
        assert tab.size == 6 * H + 16 and np.all(tab[6 * H + 4:] == 0)
        x = torch.randn(S)
        xp = np.zeros(fs.in_pad)
        xp[:S] = x.numpy()
        contract = lambda blk, v: np.array([sum(blk[f // 16, t, (f % 16) + 16 * g, e] * v[16 * t + 4 * g + e] for t in range(NT) for g in range(4)
                                                for e in range(4)) for f in range(H)])
        a0 = np.array([bias[0, f] + sum(w0[f // 16, s_, (f % 16) + 16 * g] * xp[4 * s_ + g] for s_ in range(K4) for g in range(4)) for f in range(H)])
        acts = [np.maximum(a0, 0)]
        if nh == 2:
            acts.append(np.maximum(bias[1] + contract(blocks[0], acts[0]), 0))
        else:
            assert np.all(bias[1] == 0)
        out = wh @ acts[-1] + bh
        lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
        h = x.double()
        for l, a in zip(lin[:-1], acts):
            h = torch.relu(l.weight.double() @ h + l.bias.double())
            np.testing.assert_allclose(a, h.detach().numpy(), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(out[:A], (lin[-1].weight.double() @ h + lin[-1].bias.double()).detach().numpy(), rtol=1e-12, atol=1e-12)
        assert np.all(out[A:] == 0)
        if nh == 2:
            dz = np.arange(H, dtype=np.float64) / H - 0.3
            np.testing.assert_allclose(contract(blocks[1], dz), (lin[1].weight.double().t() @ torch.from_numpy(dz)).detach().numpy(), rtol=1e-12, atol=1e-12)
    for S, A, hidden in [(5, 1, (128,) * 3), (5, 1, (64, 64)), (40, 1, (128, 128)), (5, 5, (128, 128))]:
        assert M.f32_res_supported(tg.NeuralNetwork(S, A, hidden, "ReLU")) == 0


def test_avg_reward_keeps_the_reference_list_semantics(tmp_path):
    """`Rollout_Buffer.avg_reward` is a property since the device path reads its statistic lazily; for every other use it is the
    reference's plain list (rollout_buffer.py:70, :104-121): appended by store(), replaced by load(), assignable, len() = entries."""
    g = load_golden("rollout_cartpole.npz")
    buf = tg.Rollout_Buffer(types.SimpleNamespace(env_fn=lambda: None))
    assert buf.avg_reward == []
    t = tuple(torch.from_numpy(g[f"reset_{k}"]) for k in ("obs", "act", "rew", "len", "mask"))
    buf.store(*t)
    buf.store(*t)
    assert len(buf.avg_reward) == 2 and isinstance(buf.avg_reward, list)
    buf.avg_reward = [1.0, 2.0, 3.0]
    assert buf.avg_reward == [1.0, 2.0, 3.0]
    buf.save(str(tmp_path))
    other = tg.Rollout_Buffer(types.SimpleNamespace(env_fn=lambda: None))
    assert other.load(str(tmp_path)) == 3 and other.avg_reward == [1.0, 2.0, 3.0]


def test_learner_statistics_are_a_lazily_resolved_property():
    """`last_stats` resolves a pending reader once, caches the dict and stays assignable (the learners set the reader at the end of
    learn(); nothing is read from the device until someone asks)."""
    from trajopt_grpo_amd import algorithms as ALG
    obj = ALG.GRPO.__new__(ALG.GRPO)                        # (no policy / optimizer: only the statistics bookkeeping is exercised)
    obj._stats, obj._stats_pending = {}, None
    calls = []
    obj._stats_pending = lambda: (calls.append(1), {"J": [1.0]})[1]
    assert obj.last_stats == {"J": [1.0]} and obj.last_stats == {"J": [1.0]} and calls == [1]
    obj.last_stats = {"x": 2}
    assert obj.last_stats == {"x": 2} and obj._stats_pending is None


def test_workspace_is_sized_for_its_capacity_once():
    """mlp._Workspace: a buffer is (re)allocated for max(rows, cap_rows, default_cap) rows and then only re-viewed -- the learner
    sets default_cap to its chunk size so that a growing row count never re-allocates GB-sized blocks mid-run."""
    from trajopt_grpo_amd import mlp as M
    ws = M._Workspace()
    a = ws.get("a", 10, 4, torch.float32, torch.device("cpu"))
    assert a.shape == (10, 4) and ws._buf["a"].numel() == 40
    ws.default_cap = 100
    b = ws.get("b", 10, 4, torch.float32, torch.device("cpu"))
    big = ws._buf["b"]
    assert b.shape == (10, 4) and big.numel() == 400
    c = ws.get("b", 90, 4, torch.float32, torch.device("cpu"))
    assert c.shape == (90, 4) and ws._buf["b"] is big
    d = ws.get("b", 120, 4, torch.float32, torch.device("cpu"), cap_rows=50)
    assert d.shape == (120, 4) and ws._buf["b"].numel() == 480


def test_derived_layout_keys_follow_torch_version_counters_and_raw_writes():
    """The key every derived weight layout is stamped with (mlp.GemmMLP._key / RegisterStreamF32._key): changes with any in-place
    write through torch, with a re-bound parameter and with _native.RAW_PARAM_WRITES (the fused Adam launch), and with nothing else."""
    from trajopt_grpo_amd import _native as N
    from trajopt_grpo_amd import mlp as M
    net = tg.NeuralNetwork(5, 1, (128, 128), "ReLU")
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    key = lambda: (tuple((p.data_ptr(), p._version) for l in lin for p in (l.weight, l.bias)), N.RAW_PARAM_WRITES[0])
    obj = types.SimpleNamespace(linears=lin, lin=lin)
    assert M.GemmMLP._key(obj) == key() == M.RegisterStreamF32._key(obj)
    k0 = key()
    assert M.GemmMLP._key(obj) == k0                      # reading changes nothing
    with torch.no_grad():
        lin[1].bias.add_(1.0)
    k1 = M.GemmMLP._key(obj)
    assert k1 != k0
    N.RAW_PARAM_WRITES[0] += 1
    k2 = M.GemmMLP._key(obj)
    assert k2 != k1
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    assert M.GemmMLP._key(obj) != k2


def test_ones_column_mark_follows_the_prepared_bytes_not_their_address():
    """"this input has the ones column" is marked on the STORAGE prepare_input() wrote, as a byte range.
    Slices of whole rows see it; a partial-row view, a caller's own buffer and -- the failure the old address list had -- a NEW
    allocation at a recycled address do not; a row copy (PPO's minibatch index_select) inherits it explicitly."""
    import gc
    from trajopt_grpo_amd import mlp as M
    xp = torch.zeros(64, 32, dtype=torch.bfloat16)
    assert not M.has_ones_column(xp)
    M._mark_ones_column(xp)
    assert M.has_ones_column(xp) and M.has_ones_column(xp[8:24]) and M.has_ones_column(xp[63:])
    assert not M.has_ones_column(xp[:, :16]) and not M.has_ones_column(xp.view(-1, 16)[1:3])      # not whole rows
    assert not M.has_ones_column(xp.view(32, 64))                                                    # another row length
    own = torch.zeros(64, 32, dtype=torch.bfloat16)
    assert not M.has_ones_column(own)
    # a view over a larger workspace buffer: only the prepared rows count
    big = torch.zeros(128 * 32, dtype=torch.bfloat16)
    part = big[:40 * 32].view(40, 32)
    M._mark_ones_column(part)
    assert M.has_ones_column(big[:10 * 32].view(10, 32)) and not M.has_ones_column(big[: 50 * 32].view(50, 32))
    # the mark dies with the allocation: whatever lands on the recycled address starts unmarked
    addr = xp.data_ptr()
    del xp
    gc.collect()
    for _ in range(64):
        y = torch.zeros(64, 32, dtype=torch.bfloat16)
        assert not M.has_ones_column(y)
        if y.data_ptr() == addr:
            break
    # row copies
    mb = part.index_select(0, torch.tensor([3, 1, 7]))
    assert not M.has_ones_column(mb)
    assert M.has_ones_column(M.inherit_ones_column(mb, part))
    assert not M.has_ones_column(M.inherit_ones_column(own.index_select(0, torch.tensor([0, 1])), own))
