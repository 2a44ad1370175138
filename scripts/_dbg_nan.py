import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch as th
from dualmessagepassing_amd import fused, ops, dmpnn, harness
from dualmessagepassing_amd.basemodel import build_model
from dualmessagepassing_amd.dp import FlatGradSync
from conftest import golden_files, load_golden
import test_gpu_pins_r2 as T
import tempfile, pathlib
fused.POISON_DEAD_ROWS = True
fused.USE_NODE_ROWS = bool(int(os.environ.get("NR", "0")))
gpu = th.device("cuda:0")
d = load_golden(golden_files("train_run_default")[0])
config = json.loads(str(d["config_json"]))
tmp = pathlib.Path(tempfile.mkdtemp())
train_set = T._dataset_from_fixture(d, "train", tmp)
model = build_model(json.loads(str(d["model_config_json"])), init_neigenv=float(d["init_neigenv"]), init_eeigenv=float(d["init_eeigenv"]))
model.load_state_dict({k[4:]: T._t(v) for k, v in d.items() if k.startswith("sd0.")}, strict=True)
model.to(gpu)
names = ["index","coef","residual","x","z","v_gate","e_gate","Bn","bn","Wx","Wes","be","nW2","nb2","eW2","eb2","WesT","nW2t","eW2t","slope","vpool","epool","l0","W0","WV0","edge_rows","inner"]
orig_b = fused._FusedDMPLayer.backward
cnt = [0]
def wrapped(ctx, *grads):
    gin = [(i, None if g is None else bool(th.isnan(g).any())) for i, g in enumerate(grads)]
    out = orig_b(ctx, *grads)
    bad = [names[i] for i, g in enumerate(out) if th.is_tensor(g) and bool(th.isnan(g).any())]
    print("layer backward #%d inner=%d l0=%s grads_in_nan=%s  -> NaN outputs: %s" % (cnt[0], ctx.inner, ctx.l0 is not None, gin, bad), flush=True)
    cnt[0] += 1
    return out
fused._FusedDMPLayer.backward = staticmethod(wrapped)
sync = FlatGradSync(model)
opt = th.optim.AdamW(sync.params, lr=config["lr"], weight_decay=config["weight_decay"], amsgrad=True)
sched = harness.RunSchedule(config, len(train_set))
trace = []
order = d["train_orders"][0][:config["train_batch_size"]]
tr = harness.train_epoch(model, opt, train_set.subset(list(order)) if False else train_set, config["train_batch_size"], gpu, sync=sync, bp_loss=config["bp_loss"],
                         eval_metric=config["eval_metric"], max_grad_norm=config["max_grad_norm"], order=d["train_orders"][0], schedule=sched, epoch=0,
                         match_weights=config["match_weights"], trace=trace)
print([float(a) for a, _ in trace][:4])
