"""CPU: the host-side mirror of the reference interface — constructor defaults
(src/core/icnf.jl:53-103), parameter layout, method-table errors, column sharding."""
import numpy as np
import pytest
import torch


def test_constructor_defaults_follow_the_reference(pkg):
    icnf = pkg.ICNF(nvariables=1)
    assert icnf.naugments == 2                      # naugments = nvariables + 1
    assert icnf.nn.widths == [4, 16, 16, 3]         # n_in = 1+2+1, n_hidden = 4 n_in, n_out = 3
    assert [l.act_id for l in icnf.nn.layers] == [2, 2, 0]   # softplus, softplus, identity
    assert (icnf.lambda1, icnf.lambda2, icnf.lambda3) == (0.01, 0.01, 0.01)
    assert icnf.steer_rate == 0.1 and icnf.tspan == (0.0, 1.0)
    assert icnf.S == icnf.D + 3


def test_unicode_keyword_aliases(pkg):
    icnf = pkg.ICNF(nvariables=2, **{"λ₁": 0.0, "λ₂": 0.5, "λ₃": 0.0})
    assert (icnf.lambda1, icnf.lambda2, icnf.lambda3) == (0.0, 0.5, 0.0)
    with pytest.raises(TypeError):
        pkg.ICNF(nvariables=2, bogus=1)


def test_param_layout_matches_oracle(pkg, oracles):
    o64, _ = oracles
    nn = pkg.Chain(pkg.Dense(9, 64, pkg.tanh), pkg.Dense(64, 64, pkg.tanh), pkg.Dense(64, 8))
    spec = o64.make_spec(8, [64, 64])
    assert nn.param_offsets() == tuple(spec.param_offsets())
    icnf = pkg.ICNF(nvariables=8, naugments=0, nn=nn)
    ps, st = pkg.setup(torch.Generator().manual_seed(0), icnf)
    assert ps.dtype == torch.float32 and ps.shape == (spec.param_offsets()[2],) and st == {}
    w_off, b_off, _ = nn.param_offsets()
    assert torch.all(ps[b_off[0]:b_off[0] + 64] == 0)
    assert float(ps[:w_off[1]].abs().max()) <= (6.0 / (9 + 64)) ** 0.5 + 1e-6


def test_dimension_and_method_errors(pkg):
    with pytest.raises(ValueError, match="DimensionMismatch"):
        pkg.Chain(pkg.Dense(3, 8), pkg.Dense(9, 2))
    with pytest.raises(ValueError, match="DimensionMismatch"):
        pkg.ICNF(nvariables=2, naugments=0, nn=pkg.Chain(pkg.Dense(4, 8), pkg.Dense(8, 2)))
    with pytest.raises(TypeError, match="MethodError"):
        pkg.Dense(3, 3, activation=torch.relu).act_id
    with pytest.raises(TypeError, match="MethodError"):
        pkg.ICNF(nvariables=2, compute_mode=object())
    with pytest.raises(TypeError, match="no CPU fallback"):
        pkg.ICNF(nvariables=2, device="cpu")
    icnf = pkg.ICNF(nvariables=2)                                            # the reference's default: alg = VCABM(), adaptive
    assert icnf._solver() == pkg._lib.ALG_VCABM and icnf.adaptive and isinstance(icnf.sol_kwargs["alg"], pkg.VCABM)
    with pytest.raises(NotImplementedError, match="adaptive"):
        pkg.ICNF(nvariables=2, sol_kwargs=dict(alg=pkg.VCABM(), adaptive=False, dt=0.1))._solver()
    with pytest.raises(NotImplementedError):
        pkg.ICNF(nvariables=2, sol_kwargs=dict(alg=object()))._solver()
    icnf = pkg.ICNF(nvariables=2, sol_kwargs=dict(alg=pkg.Tsit5()))          # OrdinaryDiffEq default: adaptive
    assert icnf._solver() == 1 and icnf.adaptive
    icnf = pkg.ICNF(nvariables=2, sol_kwargs=dict(alg=pkg.RK4()))            # no embedded pair on this path
    with pytest.raises(NotImplementedError, match="adaptive"):
        icnf._solver()
    icnf = pkg.ICNF(nvariables=2, sol_kwargs=dict(alg=pkg.RK4(), adaptive=False, dt=1 / 40))
    assert icnf._solver() == 0 and icnf._nsteps(0.0, 1.0) == 40
    with pytest.raises(TypeError, match="MethodError"):
        pkg.inference(icnf, pkg.TrainMode(), 1, 2, 3, 4)   # unconditioned takes (xs, ps, st)


def test_reference_compute_mode_names_resolve(pkg):
    assert pkg.LuxVecJacMatrixMode is pkg.HIPVecJacMatrixMode
    assert pkg.DIJacVecMatrixMode().jacvec and not pkg.DIVecJacMatrixMode().jacvec
    assert repr(pkg.TrainMode()) == "TrainMode{true}()" and repr(pkg.TrainMode(False)) == "TrainMode{false}()"


def test_shard_columns_partition(pkg):
    for B, G in [(65536, 8), (262144, 8), (10, 4), (3, 8), (0, 2)]:
        spans = [pkg.shard_columns(B, r, G) for r in range(G)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pkg.shard_columns(8, 2, 2)


def test_reduce_loss_single_process(pkg):
    sums = torch.tensor([10.0, 2.0, 4.0, 6.0])
    out = pkg.reduce_loss(sums, 4, (0.5, 0.25, 0.0))
    assert np.isclose(float(out), (10 + 1 + 1) / 4)


def test_planar_layer_maps_onto_a_dense_chain(pkg):
    """PlanarLayer(in => out, act) = Dense(in => 1, act) -> Dense(1 => out) with a zero second bias; its
    ComponentArray order is (u, w, b) (src/layers/planar_layer.jl:36-50)."""
    nn = pkg.Chain(pkg.PlanarLayer(6, 5, pkg.tanh))
    assert nn.widths == [6, 1, 5] and [l.act_id for l in nn.layers] == [1, 0]
    w_off, b_off, n = nn.param_offsets()
    assert (w_off, b_off, n) == ([5, 0], [11, 12], 12)          # w after u; b; u; appended zero bias
    ps = torch.arange(1, 13, dtype=torch.float32)
    ext = nn.abi_params(ps)
    assert ext.shape == (17,) and torch.all(ext[12:] == 0) and torch.equal(ext[:12], ps)
    nb = pkg.Chain(pkg.PlanarLayer(6, 5, pkg.tanh, use_bias=False))
    w_off, b_off, n = nb.param_offsets()
    assert n == 11 and b_off == [16, 11] and nb.abi_params(torch.ones(11)).shape == (17,)
    icnf = pkg.ICNF(nvariables=2, nn=nn)                      # the reference's smoke-test shape: 2*2+2 => 2*2+1
    ps, st = pkg.setup(torch.Generator().manual_seed(1), icnf)
    assert ps.shape == (12,) and float(ps[11]) == 0.0
    with pytest.raises(TypeError, match="MethodError"):
        pkg.Chain(pkg.PlanarLayer(6, 5), pkg.Dense(5, 5))


def test_epoch_batches_follow_the_dataloader_contract(pkg):
    """MLUtils.DataLoader(...; shuffle = true, partial = true): every column once per epoch, batches of
    `batchsize` with a shorter last one, batchsize 0 = one full batch, new order each epoch."""
    import torch
    g = torch.Generator().manual_seed(0)
    b1 = list(pkg.epoch_batches(10, 4, g))
    assert [len(b) for b in b1] == [4, 4, 2]
    assert sorted(torch.cat(b1).tolist()) == list(range(10))
    b2 = list(pkg.epoch_batches(10, 4, g))
    assert torch.cat(b1).tolist() != torch.cat(b2).tolist()
    full = list(pkg.epoch_batches(7, 0, g))
    assert len(full) == 1 and sorted(full[0].tolist()) == list(range(7))


def test_opt_callback_prints_like_the_reference(pkg, capsys):
    cb = pkg.make_opt_callback(64)
    assert cb(1, 3.5) is False and cb(2, 3.0) is False and cb(65, 2.5) is False
    out = capsys.readouterr().out.splitlines()
    assert out == ["Iteration: 1 | Loss: 3.5", "Iteration: 65 | Loss: 2.5"]


@pytest.mark.parametrize("kind", ["default", "planar_cond", "fixed_jvp"])
def test_machine_file_round_trip(kind, pkg, tmp_path):
    """MLJBase.save(file, mach) / machine(file) of the reference's usage example: the fitted machine as one file of plain
    numbers, names and the parameter vector; loading rebuilds an equal flow and model."""
    if kind == "default":
        icnf = pkg.ICNF(nvariables=2, lambda1=0.02)
        model = pkg.ICNFModel(icnf=icnf, batchsize=128, epochs=7)
    elif kind == "planar_cond":
        icnf = pkg.ICNF(nvariables=2, nconditions=2, nn=pkg.Chain(pkg.PlanarLayer(8, 5, pkg.tanh, use_bias=False)), steer_rate=0.0)
        model = pkg.CondICNFModel(icnf=icnf, eta=0.01)
    else:
        icnf = pkg.ICNF(nvariables=3, naugments=0, autonomous=True, compute_mode=pkg.LuxJacVecMatrixMode(), nprobes=2, epsdist="rademacher",
                        nn=pkg.Chain(pkg.Dense(3, 8, pkg.tanh), pkg.Dense(8, 3)), tspan=(0.0, 2.0),
                        sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, dt=0.05))
        model = pkg.ICNFModel(icnf=icnf)
    ps, st = pkg.setup(torch.Generator().manual_seed(3), icnf)
    f = str(tmp_path / "icnf-machine.pt")
    pkg.save_machine(f, model, (ps, st))
    model2, (ps2, st2) = pkg.load_machine(f, device="cuda:0")
    assert type(model2) is type(model) and torch.equal(ps2.cpu(), ps) and st2 == st
    for name in ("batchsize", "epochs", "weight_decay", "eta", "beta", "epsilon"):
        assert getattr(model2, name) == getattr(model, name), name
    a, b = model.icnf, model2.icnf
    for name in ("nvariables", "naugments", "nconditions", "autonomous", "tspan", "steer_rate", "lambda1", "lambda2", "lambda3",
                 "nprobes", "epsdist"):
        assert getattr(a, name) == getattr(b, name), name
    assert a.nn.widths == b.nn.widths and [l.act_id for l in a.nn.layers] == [l.act_id for l in b.nn.layers]
    assert (a.nn.planar is None) == (b.nn.planar is None) and a.nn.param_offsets() == b.nn.param_offsets()
    assert a.compute_mode.jacvec == b.compute_mode.jacvec and a._solver() == b._solver() and a.adaptive == b.adaptive
    assert {k: v for k, v in a.sol_kwargs.items() if k != "alg"} == {k: v for k, v in b.sol_kwargs.items() if k != "alg"}
    with pytest.raises(NotImplementedError):
        bad = pkg.ICNF(nvariables=2, basedist=torch.distributions.Normal(0.0, 1.0))
        pkg.save_machine(f, pkg.ICNFModel(icnf=bad), (ps, st))


def test_fixed_dt_plan_is_made_from_the_float32_values_that_cross_the_abi(pkg, oracles):
    """sol_kwargs.dt = 0.1 on (0, 1): Float32(0.1) is a hair above 1/10, so a plan made in double from the Float32 value sees
    9 whole steps and a "remainder" of almost one step.  OrdinaryDiffEq's floating-point fix-up snaps that step onto t1:
    ten equal steps, the same on the host (loss_and_gradient's grid), in the oracle and in the library (fixed_dt_plan)."""
    o64, _ = oracles
    for dt in (0.1, 1.0 / 40, 0.05, 1.0 / 3):
        g = pkg.ICNF.fixed_dt_grid(0.0, 1.0, dt)
        n = round(1.0 / dt)
        assert len(g) == n + 1 and np.allclose(np.diff(g), 1.0 / n, rtol=0, atol=1e-12), (dt, g)
        assert np.allclose(g, o64.fixed_dt_grid(0.0, 1.0, dt), rtol=0, atol=0)
    # a genuine tail survives: dt = 0.3 -> 0.3, 0.6, 0.9 and a last step of 0.1 (from the Float32 value of dt)
    g = pkg.ICNF.fixed_dt_grid(0.0, 1.0, 0.3)
    assert len(g) == 5 and abs(g[1] - float(np.float32(0.3))) < 1e-15 and abs((g[4] - g[3]) - 0.1) < 1e-6
    assert g == o64.fixed_dt_grid(0.0, 1.0, 0.3)
    icnf = pkg.ICNF(nvariables=2, sol_kwargs=dict(alg=pkg.Tsit5(), adaptive=False, dt=0.1))
    assert icnf._fixed_dt() == float(np.float32(0.1))
