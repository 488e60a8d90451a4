"""Dev tool / test helper (CPU): the 64-bit DPP FMACs of po_smooth.hip are inline assembly, which the compiler's hazard recogniser does not see.  The wait states in front of
the FIRST one of a product are in the assembly text; this check compiles the file to ISA and makes sure no VALU instruction writes the DPP source register within two
instructions in front of any other v_fmac_f64_dpp (a register copy the allocator could place there would need the wait states too).
    python tools/dpp_hazard_check.py        -> prints the count, exit code 1 on a finding"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-O3 -std=c++20 --offload-arch=gfx950 -fPIC -fno-signed-zeros -fno-honor-nans -fno-strict-aliasing -Wno-unused-result -ffp-contract=fast".split()


def check(hipcc="/opt/rocm/bin/hipcc"):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "smooth.s")
        subprocess.run([hipcc, *FLAGS, "-S", "--cuda-device-only", "-o", out, "po_smooth.hip"], cwd=os.path.join(ROOT, "path_optimizer_amd", "csrc"), check=True, capture_output=True)
        lines = open(out).read().split("\n")
    ins = [l.strip() for l in lines if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    total, findings = 0, []
    for k, l in enumerate(ins):
        if not l.startswith("v_fmac_f64_dpp"):
            continue
        total += 1
        m = re.match(r"v_fmac_f64_dpp v\[\d+:\d+\], v\[(\d+):(\d+)\]", l)
        src = (int(m.group(1)), int(m.group(2)))
        for back in (1, 2):
            p = ins[k - back]
            if p.startswith("s_nop"):
                break
            mm = re.match(r"v_\w+ (?:v\[(\d+):(\d+)\]|v(\d+))", p)
            if mm and not p.startswith("v_fmac_f64_dpp"):
                w = (int(mm.group(1)), int(mm.group(2))) if mm.group(1) else (int(mm.group(3)), int(mm.group(3)))
                if not (w[1] < src[0] or w[0] > src[1]):
                    findings.append((l, p))
    #  every PRODUCT — W = 9 DPP FMACs into one accumulator — starts with the FMAC that carries the wait states: as many `s_nop 4`-led FMACs as products
    led = sum(1 for k, l in enumerate(ins) if l.startswith("v_fmac_f64_dpp") and ins[k - 1].startswith("s_nop 4"))
    check.led = led
    return total, findings


if __name__ == "__main__":
    total, findings = check()
    print("v_fmac_f64_dpp:", total, "unprotected:", len(findings), "products led by s_nop 4:", check.led, "of", total // 9)
    if check.led * 9 != total:
        sys.exit(1)
    for f in findings:
        print("  ", f)
    sys.exit(1 if findings or total == 0 else 0)
