"""Dev tool: registers / spills / scratch / LDS per kernel from a -Rpass-analysis=kernel-resource-usage log.
usage: python tools/usage_report.py graph_learning_amd/csrc/build/gl_tile.usage.txt"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
for b in blocks:
    name = b.split("\n")[0].strip()
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    name = name.replace("(anonymous namespace)::", "").replace("(TileDev)", "")
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    print(f"{name[:70]:70s} vgpr {g('VGPRs'):3d} agpr {g('AGPRs'):3d} spill {g('VGPR Spill'):3d} scratch {g('ScratchSize .bytes/lane.'):4d} "
          f"sgpr {g('SGPRs'):3d} lds {g('LDS Size .bytes/block.')}")
