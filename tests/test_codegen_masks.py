"""Static regression test of a compiler bug that cost round 3 a "memory fault that vanishes with a clamp".

hipcc (ROCm 7.2, gfx950) compiles  base[3 * (int)(key & 0xFFFFFF)]  -- the point index of a sorted tile key, used in a
64-bit address -- as a 24-bit multiplication whose operand mask it then deletes ("a 24-bit multiply only reads the low 24
bits") and finally selects the FULL 32-bit multiply-add for:

    global_load_dword v6, v[8:9], off                   ; low half of the key: {Morton low byte | 24-bit index}
    v_mad_u64_u32 v[12:13], s[18:19], v6, 24, s[20:21]  ; frame + 24 * v6: up to 96 GB beyond the cloud

(found with rocgdb on the faulting wave, profiles/r05_g_gdb_first.txt; the source now launders the masked value through an
empty asm statement, kicp_search.hpp: key_index).  This test disassembles every gfx950 code object of the BUILT libkicp.so
and fails if a word that has just been loaded from memory reaches a constant multiply-add of an address without having
been masked, shifted or otherwise narrowed first -- the shape of that miscompile, wherever it may come back.
CPU only: llvm-objdump is part of the image, no GPU is touched."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kiss-icp_amd", "csrc", "libkicp.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def _device_listings():
    if not os.path.exists(OBJDUMP):
        pytest.skip("no llvm-objdump")
    subprocess.check_call(["make", "-C", os.path.dirname(LIB)], stdout=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="kicp_codegen_")
    try:
        shutil.copy(LIB, os.path.join(tmp, "libkicp.so"))
        subprocess.run([OBJDUMP, "--offloading", "libkicp.so"], cwd=tmp, capture_output=True, text=True, check=True)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" in name and "gfx950" in name:
                r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", name], cwd=tmp, capture_output=True, text=True, check=True)
                yield name, r.stdout.splitlines()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def find_unmasked(lines, window=14):
    """(function, defining instruction, multiply-add) for every v_mad_u64_u32 with a CONSTANT multiplier whose 32-bit operand
    was last written by a plain memory load"""
    func, hits, total = None, [], 0
    for i, line in enumerate(lines):
        m = re.match(r"[0-9a-f]+ <(.*)>:", line)
        if m:
            func = m.group(1)
            continue
        code = line.split("//")[0]
        mm = re.search(r"v_mad_u64_u32 v\[\d+:\d+\], s\[\d+:\d+\], (v\d+), (\d+|0x[0-9a-f]+),", code)
        if not mm:
            continue
        total += 1
        reg = mm.group(1)
        for k in range(i - 1, max(0, i - window), -1):
            prev = lines[k].split("//")[0]
            if re.search(r"\b%s\b" % reg, prev):  # the nearest instruction that mentions the operand
                if re.search(r"\b(global|buffer|flat|scratch)_load_dword\b", prev) and re.search(r"load_dword\s+%s\b" % reg, prev):
                    hits.append((func, prev.strip(), code.strip()))
                break
    return hits, total


def test_the_checker_recognises_the_miscompile():
    bad = """0000000000004900 <_ZN4kicp5k_icpILb0ELb0EEEvNS_9IcpParamsE>:
	global_load_dword v6, v[8:9], off                          // 00000000547C:
	s_mov_b32 s22, 0x1fffff                                    // 000000005484:
	s_waitcnt vmcnt(0)                                         // 00000000548C:
	v_mad_u64_u32 v[12:13], s[18:19], v6, 24, s[20:21]         // 000000005490:
	global_load_dword v2, v[4:5], off                          // 0000000054A0:
	v_and_b32_e32 v2, 0xffffff, v2                             // 0000000054A4:
	v_mad_u64_u32 v[2:3], s[6:7], v2, 24, v[26:27]             // 0000000054A8:
""".splitlines()
    hits, total = find_unmasked(bad)
    assert total == 2 and len(hits) == 1 and "v6" in hits[0][1]


def test_no_loaded_word_reaches_an_address_multiply_unmasked():
    seen = 0
    for name, lines in _device_listings():
        hits, total = find_unmasked(lines)
        seen += total
        assert not hits, "%s: a freshly loaded word is multiplied into an address without its mask (the ROCm 7.2 MUL_U24 miscompile):\n%s" % (
            name, "\n".join("  %s:  %s  ->  %s" % h for h in hits))
    assert seen > 20  # (the kernels do form such addresses: the scan looked at the right thing)


def _kernel_notes():
    """{kernel symbol: {metadata key: value}} of every gfx950 code object of the built library (llvm-readelf --notes)"""
    if not (os.path.exists(OBJDUMP) and os.path.exists(READELF)):
        pytest.skip("no llvm-objdump / llvm-readelf")
    subprocess.check_call(["make", "-C", os.path.dirname(LIB)], stdout=subprocess.DEVNULL)
    tmp = tempfile.mkdtemp(prefix="kicp_notes_")
    out = {}
    try:
        shutil.copy(LIB, os.path.join(tmp, "libkicp.so"))
        subprocess.run([OBJDUMP, "--offloading", "libkicp.so"], cwd=tmp, capture_output=True, text=True, check=True)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" in name and "gfx950" in name:
                r = subprocess.run([READELF, "--notes", name], cwd=tmp, capture_output=True, text=True, check=True)
                cur = None
                for line in r.stdout.splitlines():
                    m = re.match(r"\s*(?:- )?\.(\w+):\s+(\S+)", line)
                    if not m:
                        continue
                    key, val = m.group(1), m.group(2)
                    if line.lstrip().startswith("- ."):  # first key of a kernel's record
                        cur = {}
                    if cur is None:
                        continue
                    cur[key] = val
                    if key == "name":
                        out[val] = cur
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def test_the_registration_kernels_do_not_spill():
    """Round 5's thread-per-query kernel spilled 193 registers (380 bytes of scratch per lane, 335 MB of writes per launch of the
    1M-point configuration).  Both release forms of k_icp must fit their 256 registers: no spilled vector register, no scratch."""
    notes = _kernel_notes()
    for sym, form in (("_ZN4kicp5k_icpILb0ELb0EEEvNS_9IcpParamsE", "group"), ("_ZN4kicp5k_icpILb0ELb1EEEvNS_9IcpParamsE", "thread per query")):
        assert sym in notes, "k_icp (%s form) not found in the code objects: %s" % (form, sorted(k for k in notes if "k_icp" in k))
        k = notes[sym]
        assert int(k["vgpr_spill_count"]) == 0, "k_icp (%s form) spills %s vector registers" % (form, k["vgpr_spill_count"])
        assert int(k["private_segment_fixed_size"]) == 0, "k_icp (%s form) uses %s bytes of scratch per lane" % (form, k["private_segment_fixed_size"])
        assert k.get("uses_dynamic_stack", "false") == "false"
        assert int(k["vgpr_count"]) <= 256
