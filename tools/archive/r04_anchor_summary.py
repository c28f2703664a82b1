#!/usr/bin/env python3
"""profiles/r04_published_rmse.json from the logs of tools/r04_anchor.sh (gpurun_out/published_rmse_r04/eval_*.log): one record per
trained network (best checkpoint), grouped by hidden width and spectral-weight initialisation, with mean and spread per group.

    python tools/r04_anchor_summary.py
"""
import glob
import json
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "gpurun_out", "published_rmse_r04")
ROUND3 = {8: 0.01495, 27: 0.00804}        # profiles/r03_published_rmse.json (seed 1234, alpha 2.5, round-3 initialisation)
PUBLISHED = {8: 0.0139, 27: 0.0055, 38: 0.0046}


def init_of(name, rec):
    # "_cn" / "_sinit0.707": complex-normal draw (std / sqrt 2 per part), the engine default since round 4; plain names of this
    # round were trained BEFORE the change (tools/published_rmse.py header)
    if "_cn" in name or "_sinit0.707" in name:
        return "complex-normal (std / sqrt 2 per part; engine default since round 4)"
    return "round-3 draw (full std per part)"


def main():
    runs = []
    for f in sorted(glob.glob(os.path.join(SRC, "eval_*.log"))):
        for line in open(f):
            if '"what": "test RMSE' in line and "best checkpoint" in line:
                rec = json.loads(line)
                name = re.search(r"test RMSE of (\S+)", rec["what"]).group(1)
                runs.append({"name": name, "hidden": rec["hidden"], "seed": rec["seed"], "alpha": rec["alpha"],
                             "init": init_of(name, rec), "rmse_closed_loop": rec["rmse_cl"], "rmse_teacher_forced": rec["rmse_tf"],
                             "published": rec["published_rmse"], "ratio": round(rec["rmse_cl"] / rec["published_rmse"], 3),
                             "checkpoint": rec["what"].split("(")[1].rstrip(")")})
    for h, v in ROUND3.items():
        runs.append({"name": f"tfno2d64_d{h}_12-12_l4_sl50_tf10_cl40_noise0 (round 3, profiles/r03_published_rmse.json)", "hidden": h,
                     "seed": 1234, "alpha": 2.5, "init": "round-3 draw (full std per part)", "rmse_closed_loop": v,
                     "published": PUBLISHED[h], "ratio": round(v / PUBLISHED[h], 3)})
    groups = {}
    for r in runs:
        groups.setdefault((r["hidden"], r["alpha"], r["init"]), []).append(r["rmse_closed_loop"])
    summary = []
    for (h, alpha, init), vals in sorted(groups.items()):
        mean = sum(vals) / len(vals)
        summary.append({"hidden": h, "grf_alpha": alpha, "init": init, "n_runs": len(vals), "mean_rmse_closed_loop": round(mean, 5),
                        "min": min(vals), "max": max(vals), "spread_rel": round((max(vals) - min(vals)) / mean, 3),
                        "published": PUBLISHED[h], "ratio_of_mean": round(mean / PUBLISHED[h], 3)})
    note = ("tools/published_rmse.py on one MI355X (round 4, tools/r04_anchor.sh): nsbench TFNO2D trained with the published command "
            "(train_commands.txt:84-85: 500 epochs, batch 4, sequence length 50, teacher forcing 10, lr 1e-3 cosine, no clipping, no "
            "noise) on generated Navier-Stokes data (GRF alpha 2.5), evaluated closed loop on the generated test split; published "
            "numbers: src/nsbench/scripts/plot_results.py:76,82.  Findings: (1) hidden 27 with seeds 1 / 1234 under the round-3 "
            "initialisation: 0.00793 / 0.00804 -- the 1.44-1.46x is not seed scatter; (2) drawing the spectral weights as a complex "
            "normal (std / sqrt 2 per real / imaginary part instead of the full std per part; adopted as the engine's initialisation, "
            "fno_engine.py) gives hidden 27: 0.00731 / 0.00784 / 0.00812 (seeds 1234 / 1 / 2; mean 1.41x -- against the old draw's 1.45x the effect is "
            "inside the +-5 % seed scatter), hidden 8: 0.01445 / 0.0152 / 0.0145 = 1.04 / 1.09 / 1.04x over seeds 1234 / 1 / 2 (round 3: 1.08x); (3) hidden "
            "38: 0.00726 = 1.58x -- in this build the closed-loop error stops improving near 0.0073 from hidden 27 on while the "
            "published numbers keep falling (0.0055 -> 0.0046): the gap GROWS with width, so it is systematic (an optimisation or "
            "initialisation detail of the third-party FNO that matters for the wider models, or the data generator's spectrum), not "
            "noise, and it is not explained; the published sweep is itself non-monotonic at larger widths (0.0043 -> 0.0054 -> 0.0041 "
            "over 2 M -> 4 M -> 8 M parameters), about +-15 % run to run; (4) GRF alpha 2.0 instead of 2.5 (hidden 27): 0.01153 = 2.1x -- alpha "
            "2.5, which also reproduces the published persistence RMSE, is the published setting.  Protocol lines checked against src/nsbench/scripts/train.py:66-175 and evaluate.py:61-64: CosineAnnealingLR "
            "T_max = epochs, best checkpoint chosen on the validation loss, no noise, Adam defaults, batch order reshuffled per epoch.")
    out = {"_note": note, "summary": summary, "runs": runs}
    path = os.path.join(ROOT, "profiles", "r04_published_rmse.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
