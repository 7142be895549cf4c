"""Shader clock / board power / busy of ONE GPU from its sysfs hwmon files, polled from a thread (measurement plumbing for bench.py and the probes:
nothing in the product path imports this).  The GPU is found by the PCI address torch reports for the device; if that is not available, by which
card's busy counter rises while this process runs a short load."""
import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cards(root="/sys/class/drm"):
    out = {}
    for dev in sorted(glob.glob(os.path.join(root, "card*/device"))):
        if "-" in os.path.basename(os.path.dirname(dev)):      # connectors (card0-DP-1)
            continue
        hw = glob.glob(dev + "/hwmon/hwmon*")
        if not hw:
            continue
        files = {"clock": os.path.join(hw[0], "freq1_input"), "busy": os.path.join(dev, "gpu_busy_percent")}
        for name in ("power1_average", "power1_input"):
            if os.path.exists(os.path.join(hw[0], name)):
                files["power"] = os.path.join(hw[0], name)
                break
        out[os.path.realpath(dev)] = {k: p for k, p in files.items() if os.path.exists(p)}
    return out


class GpuSensors:
    def __init__(self, device=None, root="/sys/class/drm", bdf=None):
        """device: the torch device whose GPU to watch; bdf (tests): the PCI address directly, torch is not consulted."""
        self.cards = _cards(root)
        self.card, self.how, self.files = None, "none", {}
        self._rows, self._stop, self._thr = [], False, None
        idx = None
        if bdf is None:
            import torch
            idx = device.index if device is not None and device.index is not None else torch.cuda.current_device()
            try:
                pr = torch.cuda.get_device_properties(idx)
                bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            except Exception:      # noqa: BLE001 -- an older torch without the pci_* properties
                bdf = None
        for path, files in self.cards.items():
            if bdf is not None and os.path.basename(path) == bdf:
                self.card, self.how, self.files = path, "PCI address " + bdf, files
        if self.card is None and self.cards and idx is not None:
            self._by_load(idx)

    def _by_load(self, idx):
        import torch
        x = torch.randn(8192, 8192, device=f"cuda:{idx}")
        t0 = time.time()
        best = {}
        while time.time() - t0 < 1.0:
            for _ in range(10):
                x @ x
            torch.cuda.synchronize(idx)
            for path, files in self.cards.items():
                v = _read(files.get("busy", ""))
                if v and v.isdigit():
                    best[path] = max(best.get(path, 0), int(v))
        if best:
            self.card = max(best, key=best.get)
            self.how, self.files = "busy counter under a test load", self.cards[self.card]

    @property
    def available(self):
        return "clock" in self.files

    def _poll(self):
        while not self._stop:
            self._rows.append((time.time(), {k: _read(p) for k, p in self.files.items()}))
            time.sleep(0.02)

    def start(self):
        self._rows, self._stop = [], False
        self._thr = threading.Thread(target=self._poll, daemon=True)
        self._thr.start()

    def stop(self, skip=0.0):
        """-> {clock_mhz, clock_min, clock_max, power_w, busy, samples} over the samples after the first `skip` fraction."""
        self._stop = True
        if self._thr is not None:
            self._thr.join()
        rows = self._rows[int(len(self._rows) * skip):]
        out = {"samples": len(rows)}

        def col(k, scale):
            v = []
            for _, r in rows:
                try:
                    v.append(float(r.get(k)) * scale)
                except (TypeError, ValueError):
                    pass
            return v
        c, p, b = col("clock", 1e-6), col("power", 1e-6), col("busy", 1.0)
        if c:
            out.update(clock_mhz=sum(c) / len(c), clock_min=min(c), clock_max=max(c))
        if p:
            out["power_w"] = sum(p) / len(p)
        if b:
            out["busy"] = sum(b) / len(b)
        return out
