"""CPU: the C-ABI library loads and exports every symbol include/trajopt_grpo_hip.h declares;
argument validation and host-side parameter logic work without a GPU (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import trajopt_grpo_amd as tg
from oracle import envs as E

N = tg._native
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_abi_version() -> int:
    header = open(os.path.join(REPO, "include", "trajopt_grpo_hip.h")).read()
    return int(re.search(r"#define\s+TG_ABI_VERSION\s+(\d+)", header).group(1))


def test_graft_entry_build_checks_the_current_abi():
    """__graft_entry__.build() compares the built library with the binding's ABI constant, not a literal."""
    src = open(os.path.join(REPO, "__graft_entry__.py")).read()
    assert "tg._native.ABI_VERSION" in src and not re.search(r"tg_abi_version\(\)\s*==\s*\d", src)


def test_library_exports_every_declared_symbol():
    lib = N.load()
    header = open(os.path.join(REPO, "include", "trajopt_grpo_hip.h")).read()
    declared = set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    assert lib.tg_abi_version() == N.ABI_VERSION == header_abi_version()


def test_single_hip_runtime_is_shared_with_torch():
    N.load()
    libs = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    assert len(libs) == 1, libs           # one HIP runtime: torch's and the kernels' device pointers are the same world


def test_struct_layouts_match_the_header():
    assert C.sizeof(N.EnvParams) == 4 * 4 + 8 + 12 * 8
    assert C.sizeof(N.Traj) == 6 * 8 + 8 + 4 + 4
    assert C.sizeof(N.LossArgs) == 8 * 11 + 8 * 4 + 4 * 5 + 4 + 8 * 5 + 8   # incl. 4 bytes of padding before d_grad_mean; ABI 12: d_coef appended
    assert N.LossArgs.d_coef.offset == 184 and N.ChainLoss.d_norm8.offset == C.sizeof(N.ChainLoss) - 8 == 136   # ABI 12: d_norm8 appended
    assert C.sizeof(N.DwJob) == 4 * 8 + 8 + 3 * 4 + 4 + 8 and N.DwJob.d_aux.offset == 56   # tg_dw_job (ABI 4: d_aux appended)
    assert (N.TG_DW_HH, N.TG_DW_HX, N.TG_DW_DH, N.TG_DW_HR, N.TG_DW_RH) == (0, 1, 2, 3, 4)
    assert C.sizeof(N.CompactArgs) == 160 and N.CompactArgs.d_moments.offset == 136 and N.CompactArgs.rows_cap.offset == 152   # tg_compact_args


def test_env_dims_and_default_params_follow_the_reference():
    assert N.env_dims(N.TG_ENV_CARTPOLE) == (5, 1)
    assert N.env_dims(N.TG_ENV_QUADPOLE2D) == (10, 2)
    assert N.env_dims(N.TG_ENV_QUADPOLE) == (20, 4)
    p = N.default_params(N.TG_ENV_QUADPOLE, 256)
    assert list(p.p)[:10] == [1.5, 0.5, 9.80665, 0.5, 0.4, 0.4, 0.25, 0.1, 0.5, 1.5] and p.timestep == 0.02
    p = N.default_params(N.TG_ENV_QUADPOLE2D, 0)
    assert p.max_steps == 500 and list(p.p)[:8] == [1.5, 0.5, 0.4, 0.5, 0.75, 9.80665, 2.0, 0.25]
    for ms in (10, 64, 100, 128, 256, 500):
        p = N.default_params(N.TG_ENV_CARTPOLE, ms)
        assert p.time_trunc_step == E.cartpole_time_trunc_step(ms) >= ms     # float-accumulated time clause
    # Pendulum: defaults, the time clause and the balanced-step count behind `time_balanced > 5`
    assert N.env_dims(N.TG_ENV_PENDULUM) == (3, 1)
    p = N.default_params(N.TG_ENV_PENDULUM, 0)
    assert p.max_steps == 200 and p.timestep == 0.05 and list(p.p)[:4] == [1.0, 0.5, 9.80665, 0.0]
    assert p.time_trunc_step == E.cartpole_time_trunc_step(200, 0.05) and int(p.p[4]) == E.pendulum_balance_term_steps(0.05) == 101
    p.timestep = 0.02
    N.check(N.load().tg_env_finalize_params(C.byref(p)))
    assert int(p.p[4]) == E.pendulum_balance_term_steps(0.02) == 251


def test_errors_are_status_codes_with_messages():
    lib = N.load()
    assert lib.tg_env_dims(9, None, None) == -1 and b"bad env_id" in lib.tg_last_error()
    p = N.default_params(N.TG_ENV_CARTPOLE, 10)
    # null device pointer is refused on the host, before any launch
    rc = lib.tg_env_step(C.byref(p), 0, None, 1, None, 1, None, 1, None, None, None, None, 1, None)
    assert rc == -1 and b"null pointer" in lib.tg_last_error()
    rc = lib.tg_rtg_scan(None, None, 0.5, None, 4, 4, None)
    assert rc == -1
    with pytest.raises(RuntimeError, match="failed"):
        N.check(rc, "tg_rtg_scan")


def test_env_objects_carry_the_reference_surface_without_a_gpu():
    c = tg.CartPole()
    assert (c.env_name, c.max_steps, c.timestep, c._is_3d) == ("CartPole", 500, 0.02, False)
    assert c.observation_space.shape == (5,) and c.action_space.shape == (1,)
    q = tg.QuadPole(max_steps=256)
    assert q.observation_space.shape == (20,) and q.action_space.shape == (4,) and q._is_3d is True
    assert q.hover_force == pytest.approx((1.5 + 0.5) * 9.80665 / 4)
    q2 = tg.QuadPole2D()
    assert q2.observation_space.shape == (10,) and q2.action_space.shape == (2,)
    q.mass = 2.0                                   # attributes are live parameters, like the reference's
    assert q.native_params().p[0] == 2.0
    assert issubclass(tg.QuadrotorSwarm, tg.Quadrotor)
    assert c.action_space.contains(c.action_space.sample())


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    pol = tg.GaussianActor_NeuralNetwork(5, 1, (8,), cov=0.5, device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
            tg.DeviceRollout(tg.CartPole(max_steps=8), pol, 1, 2, device="cpu")
        with pytest.raises(N.NativeLibraryError, match="no CPU fallback"):
            tg.hip_ops.rtg_scan(torch.zeros(4, 4), torch.zeros(4, 4, dtype=torch.uint8), 0.9)
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(N.NativeLibraryError, match="is missing"):
        N.load()


def test_policy_surface_and_checkpoint_formats(tmp_path):
    torch.manual_seed(0)
    a = tg.GaussianActor_NeuralNetwork(5, 1, (16, 16), cov=0.5, device="cpu")
    ac = tg.GaussianActorCritic_NeuralNetwork(20, 4, (16,), cov=[0.3, 0.2, 0.5, 0.1], device="cpu")
    assert a.metadata()["num_parameters"] == 5 * 16 + 16 + 16 * 16 + 16 + 16 + 1
    assert list(a.state_dict().keys())[0] == "network.0.weight"           # reference checkpoint key names
    assert set(ac.state_dict().keys()) == {"actor", "critic"}
    assert ac.cov.shape == (4, 4) and torch.equal(torch.diagonal(ac.cov), torch.tensor([0.3, 0.2, 0.5, 0.1]))
    act, lp, v = ac(np.zeros(20))
    assert act.dtype == np.float32 and act.shape == (4,) and lp.shape == () and v.shape == (1,)
    act, lp, v = a(np.zeros((7, 5)))
    assert act.shape == (7, 1) and lp.shape == (7,) and v is None
    lp2, ent = ac.log_prob(np.zeros((3, 20)), np.zeros((3, 4), np.float32))
    dist = torch.distributions.MultivariateNormal(ac.actor(torch.zeros(3, 20)), ac.cov)   # the reference's construction
    assert torch.allclose(lp2, dist.log_prob(torch.zeros(3, 4)), atol=1e-5) and torch.allclose(ent, dist.entropy(), atol=1e-6)
    a.save(str(tmp_path)); b = tg.GaussianActor_NeuralNetwork(5, 1, (16, 16), cov=0.5, device="cpu"); b.load(str(tmp_path))
    assert all(torch.equal(x, y) for x, y in zip(a.parameters(), b.parameters()))
    assert isinstance(torch.load(tmp_path / "policy.pt", weights_only=True), dict)
    ac.save(str(tmp_path)); sd = torch.load(tmp_path / "policy.pt", weights_only=True)
    assert set(sd) == {"actor", "critic"}
    import copy
    c = copy.deepcopy(ac)                                                 # grpo.py:48 / ppo.py:62 deep-copy the policy
    assert all(torch.equal(x, y) for x, y in zip(c.parameters(), ac.parameters()))


def test_fragment_stream_layout_is_the_mfma_a_fragment_order():
    """Host logic of the fused rollout kernel's weight stream (mlp.FragmentStream), checked on CPU: block b of a
    layer holds, for every k-step, the 64 lanes' 8-element A fragments of one 32-row output tile, with the k order
    in which an MFMA accumulator tile hands its rows to the next layer: 16*ks + 8*(j>>2) + 4*h + (j&3)."""
    torch.manual_seed(0)
    net = tg.NeuralNetwork(20, 4, (128, 128), "ReLU")
    assert tg.mlp.fused_rollout_supported(net, 20, 4) == 128
    assert tg.mlp.fused_rollout_supported(net, 40, 4) == 0 and tg.mlp.fused_rollout_supported(net, 20, 5) == 0
    assert tg.mlp.fused_rollout_supported(tg.NeuralNetwork(20, 4, (128, 256), "ReLU"), 20, 4) == 0
    fs = tg.mlp.FragmentStream(net, 128)
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    KS = 128 // 16
    n_blocks = 1 + 4 + 1                       # layer 1 (all tiles), 4 output tiles of the 128x128 layer, the head
    assert fs.stream.numel() == n_blocks * KS * 64 * 8 and fs.bias.shape == (3, 128)
    st = fs.stream.float().view(-1, 64, 8)     # [fragment][lane][j]

    def expect(W, m_pad, k_pad, mo, ks, lane, j):
        r, c = 32 * mo + (lane & 31), 16 * ks + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)
        if r >= W.shape[0] or c >= W.shape[1]:
            return 0.0
        return float(W[r, c].to(torch.bfloat16))

    rng = np.random.default_rng(0)
    # layer 1: K padded 20 -> 32 (2 k-steps), 4 output tiles packed into ONE block: fragment index = mo*2 + ks
    for _ in range(200):
        mo, ks, lane, j = rng.integers(4), rng.integers(2), rng.integers(64), rng.integers(8)
        assert float(st[mo * 2 + ks, lane, j]) == expect(lin[0].weight, 128, 32, mo, ks, lane, j)
    # hidden layer: block (1 + mo), fragment ks
    base = KS
    for _ in range(200):
        mo, ks, lane, j = rng.integers(4), rng.integers(KS), rng.integers(64), rng.integers(8)
        assert float(st[base + mo * KS + ks, lane, j]) == expect(lin[1].weight, 128, 128, mo, ks, lane, j)
    # head: 4 rows padded to 32
    base = KS + 4 * KS
    for _ in range(200):
        ks, lane, j = rng.integers(KS), rng.integers(64), rng.integers(8)
        assert float(st[base + ks, lane, j]) == expect(lin[2].weight, 32, 128, 0, ks, lane, j)
    assert torch.equal(fs.bias[2, :4], lin[2].bias.detach()) and torch.all(fs.bias[2, 4:] == 0)
    # refresh follows the master weights
    with torch.no_grad():
        lin[1].weight.add_(1.0)
    fs.refresh()
    assert float(fs.stream.float().view(-1, 64, 8)[KS + 3, 5, 2]) == expect(lin[1].weight, 128, 128, 0, 3, 5, 2)


def test_chain_fragment_stream_layout_gives_every_lane_consecutive_features():
    """Host logic of the chain kernels' weight stream (mlp.FragmentStream(layout="chain")), checked on CPU.
    The products are `v_mfma_f32_16x16x32_bf16`: a 32-feature output block is two 16-row tiles ("halves" f); accumulator
    register r of lane group g = lane >> 4 is row 4 g + r of its tile.  Fragment row i of half f carries feature
    8 (i >> 2) + 4 f + (i & 3), so register r of group g in half f is feature 8 g + 4 f + r: the lane's 2 x 4 accumulators are
    the 8 consecutive features 8 g .. 8 g + 7 and the next layer reads its k in natural order, 32 ks + 8 g + j.  The forward
    head keeps its outputs in natural order (16 f + i).  A block is stored [k-step][half][64 lanes][8 bf16]."""
    torch.manual_seed(1)
    net = tg.NeuralNetwork(20, 4, (128, 128), "ReLU")
    fs = tg.mlp.FragmentStream(net, 128, layout="chain")
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    st = fs.stream.float().view(-1, 64, 8)
    assert st.shape[0] == 4 * 2 + 4 * 8 + 8          # first layer: 4 blocks x (1 k-step x 2); hidden: 4 x (4 x 2); head: 1 x (4 x 2)

    for f in range(2):                               # the permutation is what the kernels assume
        for g in range(4):
            for r in range(4):
                i = 4 * g + r
                assert 8 * (i >> 2) + 4 * f + (i & 3) == 8 * g + 4 * f + r

    def expect(W, mo, ks, f, lane, j, natural):
        i, g = lane & 15, lane >> 4
        r = 32 * mo + (16 * f + i if natural else 8 * (i >> 2) + 4 * f + (i & 3))
        c = 32 * ks + 8 * g + j
        if r >= W.shape[0] or c >= W.shape[1]:
            return 0.0
        return float(W[r, c].to(torch.bfloat16))

    rng = np.random.default_rng(1)
    for _ in range(300):
        mo, f, lane, j = rng.integers(4), rng.integers(2), rng.integers(64), rng.integers(8)
        assert float(st[mo * 2 + f, lane, j]) == expect(lin[0].weight, mo, 0, f, lane, j, False)
    for _ in range(300):
        mo, ks, f, lane, j = rng.integers(4), rng.integers(4), rng.integers(2), rng.integers(64), rng.integers(8)
        assert float(st[8 + mo * 8 + ks * 2 + f, lane, j]) == expect(lin[1].weight, mo, ks, f, lane, j, False)
    for _ in range(300):
        ks, f, lane, j = rng.integers(4), rng.integers(2), rng.integers(64), rng.integers(8)
        assert float(st[40 + ks * 2 + f, lane, j]) == expect(lin[2].weight, 0, ks, f, lane, j, True)
    # every weight appears exactly once per layer: the stream is a permutation of the zero-padded matrices
    w1 = torch.zeros(128, 128)
    w1.copy_(lin[1].weight.detach().to(torch.bfloat16).float())
    assert torch.equal(torch.sort(st[8:40].reshape(-1))[0], torch.sort(w1.reshape(-1))[0])

    # the backward stream: head first, every matrix transposed, chain rows throughout
    bs = tg.mlp.FragmentStream(net, 128, layout="chain", transposed=True)
    bt = bs.stream.float().view(-1, 64, 8)
    assert bt.shape[0] == 4 * 2 + 4 * 8              # W_head^T: K = 32 (4 outputs, padded): 4 blocks x 2; W_1^T: 4 x (4 x 2)
    for _ in range(300):
        mo, f, lane, j = rng.integers(4), rng.integers(2), rng.integers(64), rng.integers(8)
        assert float(bt[mo * 2 + f, lane, j]) == expect(lin[2].weight.t(), mo, 0, f, lane, j, False)
    for _ in range(300):
        mo, ks, f, lane, j = rng.integers(4), rng.integers(4), rng.integers(2), rng.integers(64), rng.integers(8)
        assert float(bt[8 + mo * 8 + ks * 2 + f, lane, j]) == expect(lin[1].weight.t(), mo, ks, f, lane, j, False)


def test_register_stream_f32_layout_follows_the_lds_exchange_order():
    """Host logic of tg_fused_rollout_f32's weight registers (mlp.RegisterStreamF32), checked on CPU.  A layer's output
    goes to LDS in groups of 4 consecutive features and lane (env, kh) reads group 2q + kh back, so MFMA step 4q + j
    multiplies feature 8q + 4kh + j: register 4q + j of lane (m, kh) of wave w must hold W[32w + m][8q + 4kh + j].
    The accumulator rows of lane half h, (r&3) + 8(r>>2) + 4h, are exactly the groups 2g + h it writes (g = r>>2)."""
    torch.manual_seed(2)
    net = tg.NeuralNetwork(10, 2, (128, 128, 128), "ReLU")
    assert tg.mlp.fused_rollout_f32_supported(net, 10, 2) == 128
    assert tg.mlp.fused_rollout_f32_supported(tg.NeuralNetwork(10, 2, (256, 256), "ReLU"), 10, 2) == 0
    assert tg.mlp.fused_rollout_f32_supported(tg.NeuralNetwork(10, 2, (128,) * 5, "ReLU"), 10, 2) == 0
    assert tg.mlp.fused_rollout_f32_supported(tg.NeuralNetwork(10, 2, (64, 64), "Tanh"), 10, 2) == 0
    rs = tg.mlp.RegisterStreamF32(net, 128)
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    K1 = 16                                                     # 10 inputs rounded up to 8
    R = K1 // 2 + 2 * 64
    st = rs.stream.view(4, R, 64)
    for h in range(2):
        for r in range(16):
            assert ((r & 3) + 8 * (r >> 2) + 4 * h) // 4 == 2 * (r >> 2) + h
    rng = np.random.default_rng(2)
    for _ in range(400):
        w, lane = int(rng.integers(4)), int(rng.integers(64))
        m, kh = lane & 31, lane >> 5
        r = int(rng.integers(K1 // 2))
        c = 8 * (r >> 2) + 4 * kh + (r & 3)
        assert float(st[w, r, lane]) == (float(lin[0].weight[32 * w + m, c]) if c < 10 else 0.0)
        layer, r = int(rng.integers(2)), int(rng.integers(64))
        c = 8 * (r >> 2) + 4 * kh + (r & 3)
        assert float(st[w, K1 // 2 + layer * 64 + r, lane]) == float(lin[1 + layer].weight[32 * w + m, c])
    # every weight of an H x H layer appears exactly once
    blk = st[:, K1 // 2:K1 // 2 + 64, :].reshape(-1)
    assert torch.equal(torch.sort(blk)[0], torch.sort(lin[1].weight.detach().reshape(-1))[0])
    tab = rs.table
    assert torch.equal(tab[:128], lin[0].bias.detach()) and torch.equal(tab[256:384], lin[2].bias.detach())
    head = tab[384:384 + 4 * 128].view(4, 128)
    assert torch.equal(head[:2], lin[3].weight.detach()) and torch.all(head[2:] == 0)
    assert torch.equal(tab[896:898], lin[3].bias.detach()) and torch.all(tab[898:] == 0)
    # refresh() follows the master weights
    with torch.no_grad():
        lin[2].weight.mul_(2.0)
    old = st[1, K1 // 2 + 64 + 5, 7].item()
    rs.refresh()
    assert rs.stream.view(4, R, 64)[1, K1 // 2 + 64 + 5, 7].item() == 2.0 * old


def test_register_stream_f32_sixteen_env_layout_reproduces_the_net():
    """The 16-envs-per-workgroup form of tg_fused_rollout_f32 (v_mfma_f32_16x16x4_f32: A lane (i, g) = A[i][g], B lane (j, g) = B[g][j],
    accumulator register r of lane (j, g) = row 4 g + r), restated on the host from mlp.RegisterStreamF32(block_envs=16): wave w owns
    two 16-feature tiles; the first layer's step s contracts inputs 4 s + g; an H x H layer's step 4 q + e contracts the features
    16 q + 4 g + e a lane reads back from LDS group 4 q + g.  One env through the whole net in fp64 against torch."""
    torch.manual_seed(4)
    S, A, H = 10, 2, 128
    net = tg.NeuralNetwork(S, A, (H, H, H), "ReLU")
    rs = tg.mlp.RegisterStreamF32(net, H, 16)
    lin = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    K1 = 16
    R = K1 // 2 + 2 * (H // 2)
    st = rs.stream.double().view(H // 32, R, 64).numpy()
    tab = rs.table.double().numpy()
    x = np.zeros(K1)
    x[:S] = torch.randn(S).numpy()
    a = np.zeros(H)
    for w in range(H // 32):
        for tt in range(2):
            for i in range(16):
                f = 32 * w + 16 * tt + i
                a[f] = tab[f] + sum(st[w, tt * (K1 // 4) + s_, i + 16 * g] * x[4 * s_ + g] for s_ in range(K1 // 4) for g in range(4))
    a = np.maximum(a, 0)
    h = torch.relu(lin[0].weight.double() @ torch.from_numpy(x[:S]) + lin[0].bias.double())
    np.testing.assert_allclose(a, h.detach().numpy(), rtol=1e-12, atol=1e-12)
    for l in (1, 2):
        nxt = np.zeros(H)
        base = K1 // 2 + (l - 1) * (H // 2)
        for w in range(H // 32):
            for tt in range(2):
                for i in range(16):
                    f = 32 * w + 16 * tt + i
                    nxt[f] = tab[l * H + f] + sum(st[w, base + tt * (H // 4) + 4 * q + e, i + 16 * g] * a[16 * q + 4 * g + e]
                                                  for q in range(H // 16) for e in range(4) for g in range(4))
        a = np.maximum(nxt, 0)
        h = torch.relu(lin[l].weight.double() @ h + lin[l].bias.double())
        np.testing.assert_allclose(a, h.detach().numpy(), rtol=1e-12, atol=1e-12)
    # the same tables as the 32-env form; a different permutation of the same weights
    rs32 = tg.mlp.RegisterStreamF32(net, H)
    assert torch.equal(rs.table, rs32.table) and rs.stream.numel() == rs32.stream.numel() and not torch.equal(rs.stream, rs32.stream)
    assert torch.equal(torch.sort(rs.stream)[0], torch.sort(rs32.stream)[0])
    lib = tg._native.load()
    assert lib.tg_fused_rollout_f32_block_envs(1 << 20, 1) == 32 and lib.tg_fused_rollout_f32_block_envs(64, 32) == 32


def test_every_public_member_of_the_reference_classes_exists_on_the_drop_in():
    """tests/golden/reference_public_members.json (oracle/tools/gen_members.py: the member NAMES of the reference's classes, read off the
    imported reference) against the drop-in: each name is there (`QuadPole2D.out_of_bounds`, `Env._dynamics`,
    `_wrap_action`, `Buffer.visualize`, ... were missing)."""
    import json
    from trajopt_grpo_amd import algorithms, buffers, environments, pipelines, policies
    with open(os.path.join(REPO, "tests", "golden", "reference_public_members.json")) as f:
        ref = json.load(f)
    missing = {}
    for cls_name, names in ref["classes"].items():
        cls = next((getattr(m, cls_name) for m in (tg, environments, policies, buffers, algorithms, pipelines) if hasattr(m, cls_name)), None)
        assert cls is not None, f"class {cls_name} has no drop-in"
        gone = [n for n in names if not hasattr(cls, n)]
        if gone:
            missing[cls_name] = gone
    assert not missing, missing


def test_wrap_action_is_the_reference_line_and_has_an_exact_inverse():
    """`_wrap_action` (cartpole_env.py:48-49, quadrotor_env.py:409-413, :928) reproduces the reference's float32 controls bit for bit
    (host arithmetic, no GPU), and `_unwrap_control` -- what lets `_dynamics(state, control)` run on tg_env_step -- finds an action
    that wraps to exactly that control."""
    import json
    with open(os.path.join(REPO, "tests", "golden", "reference_public_members.json")) as f:
        calls = json.load(f)["calls"]
    for name in ("CartPole", "QuadPole", "QuadPole2D"):
        env = getattr(tg, name)()
        for c in calls[name]:
            a = np.array(c["action"], dtype=np.float32)
            u = env._wrap_action(a)
            assert str(np.asarray(u).dtype) == c["wrapped_dtype"]
            assert np.array_equal(np.asarray(u, dtype=np.float64), np.array(c["wrapped"]))
            back = env._unwrap_control(u)
            assert back.dtype == np.float32 and np.array_equal(np.asarray(env._wrap_action(back), dtype=np.float64), np.array(c["wrapped"]))
    # the predicates read state_dict on the host, like the reference's
    e3, e2 = tg.QuadPole(), tg.QuadPole2D()
    for c in calls["out_of_bounds"]:
        e3.state_dict["quadrotor"] = np.array(list(c["pos"]) + [0.0] * 10)
        e2.state_dict["quadrotor"] = np.array([c["pos"][0], c["pos"][2]] + [0.0] * 6)
        assert e3._out_of_bounds() is c["QuadPole"] and e2.out_of_bounds() is c["QuadPole2D"], c
    with pytest.raises(AttributeError, match="env"):
        tg.Quadrotor.__new__(tg.Quadrotor).reset()
    assert tg.buffers.Buffer().visualize() is None
