"""Long-form leg with the side stream at normal vs high priority (bench._leg; one process per setting)."""
import json
import sys
import types

sys.path.insert(0, "/root/repo")
import torch

import bench
from styletts2_amd import ops

prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
a = types.SimpleNamespace(eager_front=False, front_priority=-1, calibrate="on", no_autotune=False, config="ljspeech")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
orig = bench.shared_stream
bench.shared_stream = lambda d, p: ops.aux_stream(d, prio if p == 0 else p)
for i in range(2):
    r = bench._leg("longform", a, dev)
    print(json.dumps({"side_priority": prio, "ms_per_step": r["ms_per_step"], "first_chunk": r["first_chunk_latency_ms"]}))
