#!/usr/bin/env python
"""Registers, LDS and the occupancy they allow for every kernel of csrc/*.hip (compiles each file with `hipcc -S`; no GPU needed).
A kernel whose VGPR + AGPR total exceeds 256 runs ONE wave per SIMD -- that is how the attention backward lost half its occupancy
unnoticed.  usage: python tools/kernel_resources.py [substring-filter]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for src in sorted(glob.glob(os.path.join(ROOT, "temporalalignnet_amd", "csrc", "*.hip"))):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-I" + os.path.join(ROOT, "include"), "-o", tmp.name, src], check=True, stderr=subprocess.DEVNULL)
        s = open(tmp.name).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", s, re.S):
        k, body = m.group(1), m.group(2)
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        get = lambda what: int((re.search(r"\.set " + re.escape(k) + r"\." + what + r", (\d+)", s) or [0, 0])[1])
        v, a, scratch = get("num_vgpr"), get("num_agpr"), get("private_seg_size")
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        if flt and flt not in name:
            continue
        tot = max(v + a, 1)
        print(f"{os.path.basename(src):18s} {name[-64:]:64s} vgpr {v:3d} agpr {a:3d} -> {min(8, 512 // tot)} waves/SIMD, "
              f"static LDS {lds:6d} B" + (f", SCRATCH {scratch} B" if scratch else ""))
