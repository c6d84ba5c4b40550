import sys, torch
sys.path.insert(0, '.')
from oracle import ref_cpu as O
from reactranker_amd import featurization, synth, functions as Fn
from tests.test_gpu_model import make_model
from tests.test_gpu_plan import _run
H, d = 32, 4
cfg = dict(hidden_size=H, mpnn_depth=d, mpnn_diff_depth=2, ffn_depth=2, use_bias=True, task_num=1, ffn_last_layer="no_softplus", task_type=None, add_features_dim=1)
for scale in (1.0, 2.0, 6.0):
    w = synth.seeded_weights(O.model_shapes(H, d, 2, 2, 1, 1, True), 5)
    w["encoder.W_h.weight"] = w["encoder.W_h.weight"] * scale
    model = make_model(cfg, w, dropout=0.1).train()
    qb = synth.make_queries(23, 300, 2, atoms_lo=14, atoms_hi=24)
    rb, pb = featurization.BatchMolGraph(qb.r_specs, K=4), featurization.BatchMolGraph(qb.p_specs, K=4)
    res = {}
    for name, f16, en in (("f16x2", True, True), ("bf16x3", False, True), ("f32", False, False)):
        Fn.SplitGemm.f16, Fn.SplitGemm.enabled = f16, en
        res[name] = _run(model, rb, pb, qb, 99, plan=True)
    Fn.SplitGemm.f16, Fn.SplitGemm.enabled = False, True
    print("scale", scale, "out max", float(res["f32"][0].abs().max()), "loss", float(res["f32"][1].sum()))
    for k in res["f32"][2]:
        ref = res["f32"][2][k]; m = float(ref.abs().max())
        print(f"  {k:28s} max {m:.3e}  f16x2-f32 {float((res['f16x2'][2][k]-ref).abs().max())/max(m,1e-30):.2e}  bf16x3-f32 {float((res['bf16x3'][2][k]-ref).abs().max())/max(m,1e-30):.2e}")
