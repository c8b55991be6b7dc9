"""Per-kernel resource table from the remarks every object of the product library leaves at build time (circuits_amd/csrc/Makefile:
-Rpass-analysis=kernel-resource-usage -> build/<file>.ru.txt): VGPRs, AGPRs, scratch bytes per lane, occupancy, LDS.
python tools/resource_usage.py [file.ru.txt ...]  (default: every file under circuits_amd/csrc/build) -> the table on stdout
(committed per round as profiles/rNN_resource_usage.txt). tests/test_resource_usage.py reads the same files."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "circuits_amd", "csrc", "build")


def parse(path):
    """[{kernel, vgprs, agprs, scratch, occupancy, lds}] of one remarks file"""
    out, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+(.*?)(\s+\[-Rpass-analysis=kernel-resource-usage\])?$", line.rstrip())
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"kernel": t.split(":", 1)[1].strip(), "file": os.path.basename(path).replace(".ru.txt", ".hip")}
            out.append(cur)
        elif cur is not None:
            for key, pat in (("vgprs", r"^VGPRs: (\d+)"), ("agprs", r"^AGPRs: (\d+)"), ("scratch", r"^ScratchSize \[bytes/lane\]: (\d+)"),
                             ("occupancy", r"^Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"^LDS Size \[bytes/block\]: (\d+)"), ("sgprs", r"^SGPRs: (\d+)")):
                mm = re.match(pat, t)
                if mm:
                    cur[key] = int(mm.group(1))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=60)
        d = r.stdout.splitlines()
        if len(d) == len(names):
            return [re.sub(r"\(.*\)$", "", x).replace("void ", "") for x in d]
    except (OSError, subprocess.TimeoutExpired):
        pass
    return names


def table(files=None):
    files = files or sorted(glob.glob(os.path.join(BUILD, "*.ru.txt")))
    rows = [r for f in files for r in parse(f)]
    for r, n in zip(rows, demangle([r["kernel"] for r in rows])):
        r["name"] = n
    return rows


if __name__ == "__main__":
    rows = table(sys.argv[1:])
    print("# kernel resource usage, hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (bytes of scratch are per lane)")
    print("%-52s %-22s %5s %5s %8s %4s %7s" % ("kernel", "file", "VGPR", "AGPR", "scratch", "occ", "LDS"))
    for r in sorted(rows, key=lambda r: (r["file"], r["name"])):
        print("%-52s %-22s %5d %5d %8d %4d %7d" % (r["name"][:52], r["file"], r.get("vgprs", 0), r.get("agprs", 0), r.get("scratch", 0), r.get("occupancy", 0), r.get("lds", 0)))
