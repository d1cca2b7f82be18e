"""Shader clock and socket power of ONE GPU from sysfs hwmon (no root, no subprocess), sampled by a thread.
The job sees one GPU but sysfs shows every card of the host: the card is matched by PCI address."""
import glob, os, threading, time


def hwmon_of(pci_bdf: str):
    """{'freq1_input': path, 'power1_input' | 'power1_average': path, 'power1_cap': path} of the card at pci_bdf ('0000:8b:00.0'), or {}"""
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if os.path.basename(os.path.realpath(os.path.join(hw, "..", ".."))) != pci_bdf:
            continue
        out = {}
        for name in ("freq1_input", "power1_average", "power1_input", "power1_cap"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out[name] = p
        return out
    return {}


def bdf_of_torch_device(index: int = 0) -> str:
    import torch
    pr = torch.cuda.get_device_properties(index)
    return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"


def _read(path):
    try:
        return int(open(path).read().strip())
    except Exception:
        return None


class PowerSampler(threading.Thread):
    def __init__(self, paths, period_s=0.02):
        super().__init__(daemon=True)
        self.paths, self.period, self.rows, self._stop_flag = paths, period_s, [], False

    def run(self):
        pw = self.paths.get("power1_average") or self.paths.get("power1_input")
        fq = self.paths.get("freq1_input")
        while not self._stop_flag:
            self.rows.append((time.perf_counter(), _read(fq) if fq else None, _read(pw) if pw else None))
            time.sleep(self.period)

    def finish(self):
        self._stop_flag = True
        self.join()

    def summary(self, t_from=None, t_to=None):
        rows = [r for r in self.rows if (t_from is None or r[0] >= t_from) and (t_to is None or r[0] <= t_to)]
        def stats(i, scale):
            v = sorted(r[i] / scale for r in rows if r[i] is not None)
            return {"median": round(v[len(v) // 2]), "min": round(v[0]), "max": round(v[-1])} if v else None
        cap = _read(self.paths["power1_cap"]) if "power1_cap" in self.paths else None
        return {"samples": len(rows), "sclk_mhz": stats(1, 1e6), "socket_power_w": stats(2, 1e6), "power_cap_w": round(cap / 1e6) if cap else None}
