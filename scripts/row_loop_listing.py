"""profiles/<tag>_synth_tile_row_loop.txt: the row loop of the SHIPPED default kernels, disassembled.

Pulls the gfx950 code objects out of multi-sdr-gps-sim_amd/gpsiq/libgpsiq.so (clang offload bundles in .hip_fatbin), runs
llvm-objdump -d on them, finds synth_tile<1, 16, 64, 1, true, 8, false> (int8) and synth_tile<2, ...> (int16), and prints
the innermost loop that holds the per-channel core (the ds_read_b32 gathers): every instruction with the issue cost of its
encoding class as measured on this chip (profiles/r01_ubench_valu_encodings.txt), and the totals the roofline discussion in
DESIGN.md section 4 quotes (VALU per 16-channel row, LDS gathers, issue cycles).  No GPU needed.
usage: python scripts/row_loop_listing.py [tag]"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "gpsiq", "libgpsiq.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CXXFILT = "c++filt"

# issue cycles per wave instruction, measured (profiles/r01_ubench_valu_encodings.txt): plain VOP2 with VGPR operands 2.5;
# SGPR-operand / VOP3 / SDWA / packed forms 4.3; v_lshl_add_u64 4.4
def issue_cycles(op, text):
    if not op.startswith("v_"):
        return None
    if op == "v_lshl_add_u64":
        return 4.4
    if "sdwa" in op or op.endswith("_e64") or op.startswith("v_pk_") or op in ("v_add3_u32", "v_lshl_add_u32", "v_perm_b32", "v_and_or_b32", "v_bfe_u32", "v_alignbit_b32", "v_mad_u32_u24"):
        return 4.3
    if re.search(r"\bs\d+|\bs\[\d+:\d+\]|\bvcc|\bexec", text.split(None, 1)[1] if " " in text else ""):
        return 4.3
    return 2.5


def code_objects():
    blob = open(LIB, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + 24)
        p = base + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[base + off: base + off + size])
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    dst = os.path.join(ROOT, "profiles", f"{tag}_synth_tile_row_loop.txt")
    lines_out = []
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(code_objects()):
            path = os.path.join(td, f"co{k}.o")
            open(path, "wb").write(co)
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
            funcs = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:\n)", dis)
            for f in funcs:
                head = f.split("\n", 1)[0]
                m = re.match(r"[0-9a-f]+ <([^>]+)>:", head)
                if not m:
                    continue
                name = subprocess.run([CXXFILT, m.group(1)], capture_output=True, text=True).stdout.strip()
                if not re.search(r"synth_tile<[12], 16, 64, 1, true, 8, false>", name):
                    continue
                body = [ln.strip() for ln in f.split("\n")[1:] if ln.strip()]
                ins = []
                for ln in body:
                    t = ln.split("//")[0].strip()
                    addr = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
                    if t:
                        ins.append((int(addr.group(1), 16) if addr else None, t))
                # innermost backward branch whose body holds the LDS gathers
                best = None
                for i, (a, t) in enumerate(ins):
                    mm = re.match(r"s_cbranch_\w+\s+(\d+)", t) or re.match(r"s_branch\s+(\d+)", t)
                    if not mm or a is None:
                        continue
                    tgt = a + 4 + 4 * (int(mm.group(1)) - (1 << 16 if int(mm.group(1)) >= 1 << 15 else 0))
                    j = next((q for q, (aa, _) in enumerate(ins) if aa == tgt), None)
                    if j is None or j > i:
                        continue
                    seg = ins[j:i + 1]
                    gathers = sum(1 for _, x in seg if x.startswith("ds_read_b32"))
                    if gathers >= 16 and (best is None or len(seg) < len(best)):
                        best = seg
                lines_out.append(f"== {name}")
                if best is None:
                    lines_out.append("   (row loop not found: the loop heuristics want updating)")
                    continue
                cls = collections.Counter()
                cyc = 0.0
                for a, t in best:
                    op = t.split()[0]
                    c = issue_cycles(op, t)
                    kind = "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_")) else "SALU/other"
                    cls[kind] += 1
                    if c:
                        cyc += c
                    lines_out.append(f"   {a:08x}  {t:<64s} {'' if c is None else f'{c:.1f}'}")
                gathers = sum(1 for _, x in best if x.startswith("ds_read_b32"))
                wide = sum(1 for _, x in best if x.startswith("ds_read_b128"))
                lines_out.append(f"   -- one pass of the loop = one 64-sample row of 16 channels: {cls['VALU']} VALU ({cls['VALU'] / 16:.2f} per channel-row), "
                                 f"{gathers} ds_read_b32 gathers + {wide} ds_read_b128, {cls['VMEM']} store(s), {cls['SALU/other']} scalar / wait / branch; "
                                 f"VALU issue at the measured rates: {cyc:.1f} cycles per row = {cyc / 16:.2f} per channel-row")
    hdr = [f"# row loop of the default kernels in the shipped libgpsiq.so (kernel id {kernels_id()}), llvm-objdump -d; right column: issue cycles of the",
           "# instruction's encoding class as measured on MI355X (profiles/r01_ubench_valu_encodings.txt: VOP2 on VGPRs 2.5, SGPR-operand / VOP3 /",
           "# SDWA / packed 4.3, v_lshl_add_u64 4.4).  Regenerate: python scripts/row_loop_listing.py <tag>", ""]
    open(dst, "w").write("\n".join(hdr + lines_out) + "\n")
    print(dst, len(lines_out), "lines")
    for ln in lines_out:
        if ln.startswith("==") or ln.startswith("   --"):
            print(ln)


def kernels_id():
    sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
    import gpsiq
    return gpsiq.kernels_id()


if __name__ == "__main__":
    main()
