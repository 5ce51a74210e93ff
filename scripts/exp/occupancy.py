"""Blocks per CU of every kernel in the built objects, by registers and by LDS (gfx950: 512 VGPRs per SIMD lane, 160 KB LDS, 4 SIMDs):
flags the kernels whose register count, not their LDS, sets the residency.  usage: python scripts/exp/occupancy.py [pattern]"""
import glob, os, re, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
rows = []
for obj in sorted(glob.glob(os.path.join(ROOT, "bert_multimodal_transformer_amd", "lib", "obj", "*.o"))):
    d = tempfile.mkdtemp()
    shutil.copy(obj, os.path.join(d, "x.o"))
    subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "x.o"], cwd=d, capture_output=True)
    dev = [f for f in os.listdir(d) if "gfx950" in f]
    if dev:
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", dev[0]], cwd=d, capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s+-?\s*\.(name|vgpr_count|vgpr_spill_count|group_segment_fixed_size|max_flat_workgroup_size|wavefront_size):\s+(\S+)", line)
            if m:
                cur[m.group(1)] = m.group(2)
                if m.group(1) == "wavefront_size":
                    rows.append((os.path.basename(obj), dict(cur)))
                    cur = {}
    shutil.rmtree(d)
print("%-14s %-86s %5s %5s %7s %6s %6s %5s" % ("object", "kernel", "thr", "vgpr", "lds", "by_reg", "by_lds", "spill"))
for obj, k in rows:
    name = k.get("name", "?")
    if pat and not pat.search(name):
        continue
    thr, vg, lds = int(k.get("max_flat_workgroup_size", 256)), int(k.get("vgpr_count", 0)), int(k.get("group_segment_fixed_size", 0))
    wps = max(1, (thr + 255) // 256)                       # waves per SIMD of one block
    alloc = max(8, (vg + 7) // 8 * 8)
    by_reg = min(8, 512 // alloc) // wps
    by_lds = (160 * 1024) // lds if lds else 99
    flag = " <-- registers" if by_reg < by_lds and by_reg < 8 else ""
    print("%-14s %-86s %5d %5d %7d %6d %6s %5s%s" % (obj, name[:86], thr, vg, lds, by_reg, by_lds if lds else "-", k.get("vgpr_spill_count", "0"), flag))
