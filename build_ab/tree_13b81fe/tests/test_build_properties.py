"""Code-generation properties the measured performance depends on (no GPU needed: hipcc cross-compiles).

The sweep kernels are bound by vector-ALU cycles and need all 8 wave slots of a SIMD: 64 VGPRs is the limit.
A behaviour-neutral source change once compiled the looping kernel to 66 VGPRs and cost 3 % (DESIGN.md
section 4); this test makes such a drift visible on the CPU box."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "isingmc.s"
    src = os.path.join(ROOT, "pyisingmontecarlo_amd", "csrc", "isingmc.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-o", str(out), src], stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernel_meta(asm, mangled_prefix):
    metas = []
    for m in re.finditer(r"\.name:\s+(" + re.escape(mangled_prefix) + r"\w*)", asm):
        after = asm[m.start():m.start() + 900]
        before = asm[max(0, m.start() - 700):m.start() + 900]
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", after).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", after).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", before).group(1))
        metas.append((m.group(1), vgpr, spill, scratch))
    return metas


@pytest.mark.parametrize("prefix", ["_ZN7isingmc21lat_sweep_loop_kernel", "_ZN7isingmc16lat_sweep_kernel"])
def test_sweep_kernels_keep_eight_waves_per_simd(device_asm, prefix):
    metas = _kernel_meta(device_asm, prefix)
    assert metas, "kernel not found in the device assembly"
    for name, vgpr, spill, scratch in metas:
        if prefix.endswith("loop_kernel") and "ILb1E" in name:
            continue  # the +-J looping instantiation needs 66-68 registers (7 waves); measured faster than the alternatives
        assert vgpr <= 64, f"{name}: {vgpr} VGPRs (> 64: fewer than 8 waves per SIMD)"
        assert spill == 0 and scratch == 0, f"{name}: spills to scratch"


def test_packed_sweep_kernel_keeps_eight_waves(device_asm):
    """Capped at 64 VGPRs with __launch_bounds__(256, 8): three spilled registers cost less than the eighth wave
    gains (the kernel is bound by the latency of its dependent memory phases): +2 % on 256^3 x 64."""
    for name, vgpr, spill, scratch in _kernel_meta(device_asm, "_ZN7isingmc15pk_sweep_kernel"):
        assert vgpr <= 64 and spill <= 4 and scratch <= 32, f"{name}: {vgpr} VGPRs, {spill} spills, {scratch} B scratch"


def test_packed_uniform_degree_kernels_keep_eight_waves(tmp_path):
    """pk_sweep_uni_kernel<D, UB, PMJ, TABLE>: at most a few spilled registers at __launch_bounds__(256, 8)."""
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path / "pku.s"
    src = os.path.join(ROOT, "pyisingmontecarlo_amd", "csrc", "packed_uni_kernels.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-o", str(out), src], stderr=subprocess.DEVNULL)
    metas = _kernel_meta(out.read_text(), "_ZN7isingmc19pk_sweep_uni_kernel")
    assert len(metas) == 32
    for name, vgpr, spill, scratch in metas:
        assert vgpr <= 64 and spill <= 4 and scratch <= 32, f"{name}: {vgpr} VGPRs, {spill} spills, {scratch} B scratch"
        if "ELb0ELb" in name:  # one coupling sign: the c5 kernels
            assert spill == 0 and scratch == 0, f"{name}: spills"


def test_multi_class_kernels_keep_their_occupancy(tmp_path):
    """lat_mc_sweep_kernel: the uniform-field instantiation the field benchmark runs (<FIELD, uniform sign, 2^k mapping, no
    sign planes>) stays at 7 waves per SIMD (<= 72 VGPRs; a runtime branch on the sign-plane pointer once made it 97), the
    open-boundary ones at 8 (<= 64), and nothing spills."""
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path / "mc.s"
    src = os.path.join(ROOT, "pyisingmontecarlo_amd", "csrc", "mc_kernels.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-o", str(out), src], stderr=subprocess.DEVNULL)
    metas = _kernel_meta(out.read_text(), "_ZN7isingmc19lat_mc_sweep_kernel")
    assert len(metas) == 24
    for name, vgpr, spill, scratch in metas:
        assert spill == 0 and scratch == 0, f"{name}: spills"
        if "ILi1ELb0ELb1ELb0E" in name:
            assert vgpr <= 72, f"{name}: {vgpr} VGPRs"
        if "ILi2E" in name:
            assert vgpr <= 65, f"{name}: {vgpr} VGPRs"


def test_no_vector_store_data_is_overwritten_right_behind_the_store(device_asm, tmp_path):
    """A buffer_store_dwordx3/x4 reads its data registers some cycles after it issues.  The compiler's hazard model
    (GCNHazardRecognizer::createsVALUHazard) inserts wait states only when the store's soffset is an immediate; with a REGISTER
    soffset it inserts none -- and on a loaded MI355X the fused sweep+measure kernel, whose bit counts overwrote the data
    registers in the very next instruction, stored bit counts instead of spins (round 3; wrong configurations from ~1500
    workgroups per launch on; isolated in profiles/r03_store_hazard.txt: the window is the one instruction slot behind the store).
    No kernel of any translation unit may write such a store's data registers within 8 instructions of it."""
    texts = {"isingmc.hip": device_asm}
    procs = []
    for name in ("mc_kernels.hip", "strip_kernels.hip", "spread_kernels.hip", "packed_uni_kernels.hip", "real_kernels.hip"):   # every other translation unit
        out = tmp_path / (name + ".s")
        procs.append((name, out, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                                                   "--cuda-device-only", "-o", str(out), os.path.join(ROOT, "pyisingmontecarlo_amd", "csrc", name)],
                                                  stderr=subprocess.DEVNULL)))
    for name, out, proc in procs:
        assert proc.wait() == 0, name
        texts[name] = out.read_text()
    store = re.compile(r"\s(buffer_store_dwordx[34])\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(\S+)")
    write = re.compile(r"(v_\w+)\s+v(?:\[(\d+):(\d+)\]|(\d+))")
    stores = 0
    for name, text in texts.items():
        lines = text.split("\n")
        kernel = "?"
        for i, line in enumerate(lines):
            m0 = re.match(r"^(_Z\w+):", line)
            if m0:
                kernel = m0.group(1)
            m = store.search(line)
            if not m or not re.match(r"s\d+$", m.group(4)):
                continue  # immediate soffset: the compiler inserts the wait states itself
            stores += 1
            lo, hi = int(m.group(2)), int(m.group(3))
            seen, j = 0, i + 1
            while j < len(lines) and seen < 8:
                t = lines[j].strip()
                j += 1
                if not t or t.startswith((";", ".")):
                    continue
                seen += 1
                w = write.match(t)
                if w and not t.startswith(("v_cmp", "v_cmpx")):
                    a = int(w.group(2) or w.group(4))
                    b = int(w.group(3) or w.group(4))
                    assert b < lo or a > hi, f"{name}: {kernel[:70]}: `{t}` {seen} instruction(s) behind `{line.strip()}`"
    assert stores >= 4, "no register-offset vector stores found: has the pattern of this test gone stale?"
