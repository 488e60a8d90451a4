"""Dev tool (CPU): registers / spills / scratch of every kernel in the objects libpo_hip.so is LINKED from, read from the gfx950 code object's metadata notes.
The object list is the Makefile's own (`make -pn` -> the prerequisites of ../libpo_hip.so), not a glob of .build/ — a box with left-over objects of earlier rounds used to
put 131 rows of kernels that no longer ship into profiles/r5c/kernel_resources.txt.
    python tools/kernel_resources.py [substring of the object name ...]          (needs /opt/rocm/lib/llvm/bin/{llvm-objcopy,llvm-readelf})"""
import glob, os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        if subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], capture_output=True).returncode or not os.path.exists(fat):
            return []
        d = open(fat, "rb").read()
        if d[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
            return []
        n = struct.unpack("<Q", d[24:32])[0]
        p = 32
        out = []
        for _ in range(n):
            off, size, tl = struct.unpack("<QQQ", d[p:p + 24]); p += 24
            t = d[p:p + tl].decode(); p += tl
            if "gfx950" not in t or size == 0:
                continue
            co = os.path.join(td, "dev.co")
            open(co, "wb").write(d[off:off + size])
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "agpr_count" and cur.get("name"):
                    out.append(cur); cur = {}
                if k in ("name", "agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                    cur[k] = v
            if cur.get("name"):
                out.append(cur)
        return out


def linked_objects():
    """The objects of the library's link line, as make itself resolves them (basenames, in link order)."""
    csrc = os.path.join(ROOT, "path_optimizer_amd", "csrc")
    db = subprocess.run(["make", "-pn", "-C", csrc], capture_output=True, text=True).stdout
    m = re.search(r"^\.\./libpo_hip\.so:(.*)$", db, flags=re.M)
    if not m:
        raise SystemExit("kernel_resources: no rule for ../libpo_hip.so in the Makefile's database")
    return [os.path.basename(o) for o in m.group(1).split() if o.endswith(".o")]


def demangle(n):
    r = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(po::Dev\w+(, po::Dev\w+)*\)", "", r) or n


if __name__ == "__main__":
    pats = sys.argv[1:]
    print(f"{'object':28s} {'kernel':62s} {'vgpr':>5s} {'agpr':>5s} {'spillV':>6s} {'spillS':>6s} {'scratch B/lane':>14s}")
    for base in sorted(linked_objects()):
        obj = os.path.join(ROOT, "path_optimizer_amd", "csrc", ".build", base)
        if not os.path.exists(obj):
            raise SystemExit(f"kernel_resources: {obj} is on the link line but not built")
        if pats and not any(p in base for p in pats):
            continue
        for k in kernels_of(obj):
            print(f"{base:28s} {demangle(k['name'])[:62]:62s} {k.get('vgpr_count', '?'):>5s} {k.get('agpr_count', '?'):>5s} {k.get('vgpr_spill_count', '?'):>6s} {k.get('sgpr_spill_count', '?'):>6s} {k.get('private_segment_fixed_size', '?'):>14s}")
