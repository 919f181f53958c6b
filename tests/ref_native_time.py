"""Timing of the reference's own two native units next to their GPU replacements.  TEST INFRASTRUCTURE.

The compiled reference units (oracle/_ref, built from the reference's .pyx by `make -C oracle ref`) stay in the BUILD
CONTAINER: oracle/_ref/ is listed in .gpurunignore and never reaches the GPU box.  So this script has two legs that
run in different places:

    python tests/ref_native_time.py ref    # build container (needs oracle/_ref): times the reference units on the host cores;
                                           # the numbers go into BASELINE.md section 2
    python tests/ref_native_time.py gpu    # GPU box: times pysp_build_map_f32 / pysp_warp_table_f32 on the same inputs
                                           # (bit-exactness against the reference is pinned by fixtures G3 / G7)
"""
import importlib.machinery, importlib.util, os, sys, sysconfig, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W = 4000, 6000


def load_ref(base):
    path = os.path.join(ROOT, "oracle", "_ref", base + sysconfig.get_config_var("EXT_SUFFIX"))
    loader = importlib.machinery.ExtensionFileLoader(base, path)
    spec = importlib.util.spec_from_file_location(base, path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def lab_input():
    rng = np.random.default_rng(1)
    return np.ascontiguousarray(rng.random((H + 2, W + 2, 3), dtype=np.float32) * np.float32(100))


def timed(name, fn, extra=lambda: ""):
    fn(); t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name}: {dt * 1e3:.1f} ms = {H * W / 1e6 / dt:.0f} MP/s{extra()}")


def ref_leg():
    ref_map, ref_warp = load_ref("ahd_homogeneity_cython"), load_ref("dng_warp_rectilinear_coords")
    lab = lab_input()
    timed("reference build_map (Cython + OpenMP, all host cores)", lambda: ref_map.build_map(lab, 1, 3, False))
    timed("reference compute_remapping_table (Cython + OpenMP)", lambda: ref_warp.compute_remapping_table(1.0, 0.01, 0.002, 0.0, 0.0, 0.0, W, H, 0.5, 0.5, 1.0))
    print("cores:", len(os.sched_getaffinity(0)))


def gpu_leg():
    from pysp_amd import _lib
    from pysp_amd.debayer.ahd_homogeneity_cython import build_map as gpu_build_map
    from pysp_amd.dng_warp_corr.dng_warp_rectilinear_coords import compute_remapping_table as gpu_table
    lab = lab_input()
    kms = lambda: f", kernel {_lib.default_context().last_kernel_ms():.3f} ms"
    timed("pysp_build_map_f32 (incl. PCIe)", lambda: gpu_build_map(lab, 1, 3, False), kms)
    timed("pysp_warp_table_f32 (incl. PCIe)", lambda: gpu_table(1.0, 0.01, 0.002, 0.0, 0.0, 0.0, W, H, 0.5, 0.5, 1.0), kms)


if __name__ == "__main__":
    leg = sys.argv[1] if len(sys.argv) > 1 else ("ref" if os.path.isdir(os.path.join(ROOT, "oracle", "_ref")) else "gpu")
    {"ref": ref_leg, "gpu": gpu_leg}[leg]()
