import sys, json, types
sys.path.insert(0, "/root/repo")
import torch, bench
a = types.SimpleNamespace(eager_front=False, front_priority=-1, calibrate="on", no_autotune=False, config="ljspeech")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = bench._leg("libritts_hifigan", a, dev)
print(json.dumps({k: r[k] for k in ("ms_per_step", "schedule", "schedules_ms_per_step", "wall_s")}))
r = bench._leg("libritts_hifigan", a, dev)
print(json.dumps({k: r[k] for k in ("ms_per_step", "schedule", "schedules_ms_per_step", "wall_s")}))
