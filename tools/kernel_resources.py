"""Per-kernel register / scratch summary from hipcc's resource-usage remarks.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
          music-generator_amd/csrc/dj_lstm.hip -o /tmp/x.s 2> /tmp/res.log
    python tools/kernel_resources.py /tmp/res.log [substring]        # kernels with scratch are marked "SPILL"
"""
import re
import subprocess
import sys


def main():
    log = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    names = re.findall(r"Function Name: (\S+)", log)
    try:
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    except Exception:
        dem = names
    blocks = re.split(r"remark: Function Name: ", log)[1:]
    for name, d, b in zip(names, dem, blocks):
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"\(.*", "", d).replace("void ", "")
        if pat and pat not in d and pat not in name:
            continue
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        sc, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
        print(f"{d[:70]:70s} VGPR {g('VGPRs'):3d} AGPR {g('AGPRs'):3d} scratch {sc:4d} occ {occ}" + ("   SPILL" if sc else ""))


if __name__ == "__main__":
    main()
