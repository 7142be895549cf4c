"""Checkpoint streaming: the reference's `saveModel` (diff_model.py:489-536, called from model_trainer.py:545-548)
pickles the model, the EMA copy, the optimizer, the scheduler and the scaler synchronously -- the training loop stands still for
a device-to-host copy of every tensor plus the file writes (≈5 GB for MMDiT-B, 18 GB for MMDiT-L).

With 288 GB of HBM the snapshot can stay on the device: `CheckpointStreamer.save`
  1. clones every CUDA tensor of the objects to be saved on the training stream (device-to-device, a few ms; from here on the
     training step may overwrite the originals),
  2. hands the clones to a writer thread that copies them into pinned host buffers on a side stream (PCIe traffic and the
     pinned allocations overlap the following training steps) and
  3. `torch.save`s the reference's six files from the host copies.
File names, pickle contents and the JSON are exactly those of `diff_model.saveModel`.  One checkpoint is in flight at a time
(`save` waits for the previous one); `wait()` joins the writer and re-raises its error, and runs at interpreter exit.
"""
import atexit
import json
import os
import threading

import torch


def _map_tensors(obj, fn):
    """Structure-preserving copy of nested dict / list / tuple containers with fn applied to every tensor."""
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return type(obj)((k, _map_tensors(v, fn)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class CheckpointStreamer:
    def __init__(self, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self._thread, self._error = None, None
        self.saves = 0
        atexit.register(self._finish_quietly)

    # ------------------------------------------------------------------------------------------
    def _snapshot(self, obj):
        """Device tensors -> private device clones (training stream); host tensors -> private host clones."""
        return _map_tensors(obj, lambda t: t.detach().clone())

    def _to_host(self, obj):
        def mv(t):
            if not t.is_cuda:
                return t
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            return h
        return _map_tensors(obj, mv)

    def save(self, files, json_files=()):
        """files: [(object, path)] for torch.save; json_files: [(dict, path)].  Returns as soon as the snapshot is enqueued."""
        self.wait()
        snap = [(self._snapshot(o), p) for o, p in files]
        js = [(json.loads(json.dumps(d)), p) for d, p in json_files]
        ready = None
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))     # the clones are complete once this event has fired

        def write():
            try:
                host = snap
                if self.cuda:
                    # pinned allocation (slow the first time) and the PCIe copies run here, off the training thread
                    torch.cuda.set_device(self.device)
                    with torch.cuda.stream(self.stream):
                        self.stream.wait_event(ready)
                        host = [(self._to_host(o), p) for o, p in snap]
                    self.stream.synchronize()
                    snap.clear()                                     # device clones released: the copy out of them has finished
                for o, p in host:
                    os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
                    torch.save(o, p + ".tmp")
                    os.replace(p + ".tmp", p)
                for d, p in js:
                    with open(p, "w") as f:
                        json.dump(d, f)
            except BaseException as e:   # surfaced by wait()
                self._error = e

        self._thread = threading.Thread(target=write, name="checkpoint-writer", daemon=False)
        self._thread.start()
        self.saves += 1

    def wait(self):
        """Block until the checkpoint in flight is on disk; re-raise a writer error."""
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            e, self._error = self._error, None
            raise RuntimeError(f"checkpoint writer failed: {e!r}") from e

    def _finish_quietly(self):
        try:
            self.wait()
        except Exception as e:   # interpreter exit: report, do not raise
            print(f"[checkpoint_stream] {e}", flush=True)
