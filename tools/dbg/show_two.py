"""Print the interesting fields of gpurun_out/two_rank.json (a --gpus 2 bench line)."""
import json
d = json.load(open("gpurun_out/two_rank.json"))
print(d["n_gpus"], d["ms_per_step"], d["value"], d["scaling"])
print(d.get("exchange"))
print({k: d["also"][k] for k in d.get("also", {}) if k in ("ms_per_mpc_step", "value", "exchange")})
