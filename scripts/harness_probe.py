import numpy as np, torch as th, time
from dualmessagepassing_amd.basemodel import build_model
from dualmessagepassing_amd.dp import FlatGradSync
from dualmessagepassing_amd.harness import SyntheticPairs, evaluate_epoch, train_epoch
gpu = th.device("cuda:0")
for shape in ((3, 2, 8, 16, 2, 1), (3, 3, 10, 30, 1, 2), (3, 3, 10, 24, 2, 2)):
    ds = SyntheticPairs(128, *shape, seed=3)
    c = np.array([s["counts"] for s in ds.samples]); print(shape, "counts mean %.2f max %d var %.3f" % (c.mean(), c.max(), c.var()))
    for rep in ("DMPNN", "CompGCN"):
        for lr in (1e-3, 3e-3):
            th.manual_seed(0)
            model = build_model(**ds.model_config(hid_dim=32, layers=2, rep_net=rep)).to(gpu)
            sync = FlatGradSync(model)
            opt = th.optim.AdamW(sync.params, lr=lr, weight_decay=1e-5, amsgrad=True)
            b = evaluate_epoch(model, ds, 32, gpu)["MSE"]
            t = time.time()
            hist = [train_epoch(model, opt, ds, 32, gpu, sync=sync, neg_slp=0.01)["bp_loss"] for _ in range(60)]
            print(" ", rep, lr, "before %.3f after %.3f (%.1fs)" % (b, evaluate_epoch(model, ds, 32, gpu)["MSE"], time.time() - t), np.round(hist[::6], 3))
