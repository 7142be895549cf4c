"""What the flight recorder publishes about the watchdog (model_trainer._wait_for_watchdog polls it before a capture): one-rank RCCL
group, a few eager all-reduces, the status right after the device synchronize and 20 / 150 / 300 ms later."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
dist.init_process_group("nccl", rank=0, world_size=1)
import sd3_amd  # noqa: E402,F401
from sd3_amd.model_trainer import _watchdog_status  # noqa: E402

x = torch.ones(1 << 20, device="cuda")
for i in range(5):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.time()
for wait in (0.0, 0.02, 0.15, 0.3):
    time.sleep(wait)
    from torch._C import _distributed_c10d as c10d
    raw = json.loads(c10d._dump_nccl_trace_json(includeCollectives=False, onlyActive=True))
    print(f"+{time.time() - t0:.3f}s status={_watchdog_status()} raw_pg_status={raw.get('pg_status')}", flush=True)
dist.destroy_process_group()
