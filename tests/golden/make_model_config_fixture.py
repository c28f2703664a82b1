#!/usr/bin/env python3
"""Constructor kwargs of the reference's shipped model configs, as a JSON fixture.

Reads src/{nsbench,dlwpbench}/configs/model/*.yaml of the reference (a config schema is data, not code) for the model
families on the hot path, resolves the `${data.*}` interpolations against the default data group of each app
(nsbench configs/data/navier-stokes_s64.yaml: 64 x 64; dlwpbench configs/data/example.yaml: 32 x 64) and writes
tests/golden/shipped_model_configs.json.  tests/test_gpu_shipped_configs.py builds every class from these kwargs the way
train.py does (`eval(cfg.model.type)(**cfg.model)`, nsbench/scripts/train.py:66, dlwpbench/scripts/train.py:39).

    python tests/golden/make_model_config_fixture.py
"""
import json
import os

import yaml

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shipped_model_configs.json")
HOT = {"nsbench": ["fno", "fourcastnet", "swintransformer"],
       "dlwpbench": ["fno", "fourcastnet", "fourcastnetv2", "swintransformer", "panguweather", "sfno"]}
DATA = {"nsbench": {"data.height": 64, "data.width": 64}, "dlwpbench": {"data.height": 32, "data.width": 64}}


def resolve(v, env):
    if isinstance(v, str) and v.startswith("${") and v.endswith("}"):
        return env[v[2:-1]]
    return v


def main():
    out = {}
    for app, names in HOT.items():
        for name in names:
            path = f"{REF}/{app}/configs/model/{name}.yaml"
            with open(path) as f:
                cfg = yaml.safe_load(f)
            out[f"{app}/{name}"] = {"source": f"src/{app}/configs/model/{name}.yaml",
                                    "kwargs": {k: resolve(v, DATA[app]) for k, v in cfg.items()}}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, len(out), "configs")


if __name__ == "__main__":
    main()
