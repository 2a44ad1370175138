"""Config 3 at a dataset's scale (VERDICT r3 "missing" 5): the training loop on a few thousand ragged pairs of the
reference's "small" size ranges, on the GPU path and -- same initial parameters, same batches in the same order, same loss,
clipping and AdamW(amsgrad) -- on the CPU oracle of the whole model (oracle/model_oracle.py), epoch by epoch.

Not collected by pytest (minutes of host time): run it on the GPU box,
    python tests/config3_scale_run.py [pairs] [epochs] > profiles/rNN_config3_small.txt
It lives under tests/ because only tests may execute the oracle.  The published "small" dataset is a download that is not
available offline: the pairs are ``harness.SmallLikePairs`` (directed ER graphs, uniform labels, exact counts)."""
import os
import sys
import time

import numpy as np
import torch as th
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def oracle_batch(ds, idx):
    sides = {}
    for key in ("pattern", "graph"):
        off, parts = 0, {k: [] for k in ("src", "dst", "rev", "elabel", "eid", "label", "id")}
        bnn, bne = [], []
        for i in idx:
            g = ds.samples[i][key]
            parts["src"].append(g["src"] + off); parts["dst"].append(g["dst"] + off)
            parts["rev"].append(g["rev"]); parts["elabel"].append(g["elabel"]); parts["eid"].append(g["eid"])
            parts["label"].append(g["vlabel"]); parts["id"].append(np.arange(g["num_nodes"]))
            off += g["num_nodes"]
            bnn.append(int(g["num_nodes"])); bne.append(len(g["src"]))
        t = {k: th.from_numpy(np.concatenate(v)) for k, v in parts.items()}
        t["bnn"], t["bne"] = bnn, bne
        sides[key] = t
    counts = th.tensor([float(ds.samples[i]["counts"]) for i in idx]).view(-1, 1)
    return sides["pattern"], sides["graph"], counts


def main():
    import model_oracle as MO
    from dualmessagepassing_amd import harness
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    bs, lr, wd, clip, hid, slp = 64, 2e-4, 1e-5, 8.0, 64, 0.01      # slp: negative slope of the count loss (train.py:624-628; the heads' last
                                                                    # Linear starts at zero, pred_c = 0: a slope of 0 would never move it)
    gpu = th.device("cuda:0")
    t0 = time.perf_counter()
    data = harness.SmallLikePairs(pairs, seed=11)
    n_train = pairs * 4 // 5
    train, dev = data.subset(range(n_train)), data.subset(range(n_train, pairs))
    cnt = np.array([x["counts"] for x in data.samples], float)
    print("dataset: %d pairs (%d train / %d dev) generated in %.1f s; counts: %.0f %% zero, median %.0f, mean %.1f, max %.0f"
          % (pairs, n_train, pairs - n_train, time.perf_counter() - t0, 100 * (cnt == 0).mean(), np.median(cnt), cnt.mean(), cnt.max()))
    dev_cnt = np.array([x["counts"] for x in dev.samples], float)
    print("dev MAE of predicting 0: %.3f; of predicting the training mean: %.3f"
          % (np.abs(dev_cnt).mean(), np.abs(dev_cnt - cnt[:n_train].mean()).mean()))
    mc = data.model_config(hid_dim=hid, layers=3, rep_act_func="leaky_relu", pred_act_func="leaky_relu", emb_net="Equivariant",
                           share_emb_net=True, share_enc_net=True, init_neigenv=4.0, init_eeigenv=4.0)
    th.manual_seed(0)
    model = build_model(**mc).to(gpu)
    init = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    # ---- the GPU path: harness.fit (train_epoch + evaluate_epoch), eager launches (every batch is a new shape)
    sync = FlatGradSync(model)
    opt = FlatAdamW([sync.flatten_parameters()], lr=lr, weight_decay=wd, amsgrad=True)
    t0 = time.perf_counter()
    hist = harness.fit(model, opt, train, dev, epochs, bs, gpu, sync=sync, seed=3, max_grad_norm=clip, neg_slp=slp)
    th.cuda.synchronize()
    t_gpu = time.perf_counter() - t0

    # ---- the CPU oracle: the same run (fit's batch order: rng.permutation per epoch from seed 3)
    th.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = {k: v.clone() for k, v in init.items()}
    for k in list(sd):
        twin = "g_" + k[2:]
        if k.startswith("p_") and twin in sd and sd[k].shape == sd[twin].shape and th.equal(sd[k], sd[twin]):
            sd[k] = sd[twin]
    params = []
    for k, v in sd.items():
        if v.is_floating_point() and "enc_net" not in k and not any(v is q for q in params):
            params.append(v.requires_grad_(True))
    ref_opt = th.optim.AdamW(params, lr=lr, weight_decay=wd, amsgrad=True)
    rng = np.random.default_rng(3)
    t0 = time.perf_counter()
    ref = []
    for epoch in range(epochs):
        order = rng.permutation(len(train))
        tot, n = 0.0, 0
        for i in range(0, len(order), bs):
            idx = order[i:i + bs]
            pattern, graph, counts = oracle_batch(train, idx)
            pred = MO.model_forward(sd, mc, pattern, graph)["pred_c"]
            loss = F.mse_loss(F.leaky_relu(pred, slp), counts)
            ref_opt.zero_grad(set_to_none=True)
            loss.backward()
            th.nn.utils.clip_grad_norm_(params, clip)
            ref_opt.step()
            tot += float(loss.detach()) * len(idx); n += len(idx)
        with th.no_grad():
            preds, tgts = [], []
            for i in range(0, len(dev), bs):
                idx = np.arange(i, min(i + bs, len(dev)))
                pattern, graph, counts = oracle_batch(dev, idx)
                preds.append(F.relu(MO.model_forward(sd, mc, pattern, graph)["pred_c"])); tgts.append(counts)
            p, t = th.cat(preds), th.cat(tgts)
        ref.append({"train_loss": tot / n, "dev_MAE": float(F.l1_loss(p, t)), "dev_MSE": float(F.mse_loss(p, t))})
    t_cpu = time.perf_counter() - t0
    print("GPU path: %d epochs in %.1f s (%.0f pairs/s incl. evaluation and host batching); CPU oracle: %.1f s (%.0f pairs/s, %d threads)"
          % (epochs, t_gpu, epochs * pairs / t_gpu, t_cpu, epochs * pairs / t_cpu, th.get_num_threads()))
    print("epoch | train loss GPU / CPU oracle (rel. diff) | dev MAE GPU / CPU oracle (rel. diff) | dev MSE GPU / CPU")
    worst = 0.0
    for e, (h, r) in enumerate(zip(hist, ref)):
        dl = abs(h["train"]["bp_loss"] - r["train_loss"]) / max(r["train_loss"], 1e-9)
        dm = abs(h["dev"]["MAE"] - r["dev_MAE"]) / max(r["dev_MAE"], 1e-9)
        worst = max(worst, dm)
        print("%5d | %10.4f / %10.4f (%.2e) | %8.4f / %8.4f (%.2e) | %9.3f / %9.3f"
              % (e, h["train"]["bp_loss"], r["train_loss"], dl, h["dev"]["MAE"], r["dev_MAE"], dm, h["dev"]["MSE"], r["dev_MSE"]))
    k = max(1, epochs // 4)
    g_tail, c_tail = np.mean([h["dev"]["MAE"] for h in hist[-k:]]), np.mean([r["dev_MAE"] for r in ref[-k:]])
    print("final dev MAE: GPU %.4f, CPU oracle %.4f; mean of the last %d epochs: GPU %.4f, CPU oracle %.4f (rel. diff %.2e); "
          "best epoch: GPU %.4f, CPU oracle %.4f; largest relative difference of the dev MAE over the run: %.2e"
          % (hist[-1]["dev"]["MAE"], ref[-1]["dev_MAE"], k, g_tail, c_tail, abs(g_tail - c_tail) / c_tail,
             min(h["dev"]["MAE"] for h in hist), min(r["dev_MAE"] for r in ref), worst))


if __name__ == "__main__":
    main()
