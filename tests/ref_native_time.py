"""Timing of the reference's own native units (oracle/_ref, compiled from the reference's .pyx by oracle/Makefile ref) next to their
GPU replacements, on the GPU box: python tests/ref_native_time.py.  Test infrastructure (uses oracle/_ref); the only pieces of the
reference that are native code and can travel as binaries -- the rest of its path is NumPy/OpenCV and is timed as the oracle port."""
import importlib.machinery, importlib.util, os, sys, sysconfig, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_ref(base):
    path = os.path.join(ROOT, "oracle", "_ref", base + sysconfig.get_config_var("EXT_SUFFIX"))
    loader = importlib.machinery.ExtensionFileLoader(base, path)
    spec = importlib.util.spec_from_file_location(base, path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def main():
    from pysp_amd import _lib
    from pysp_amd.debayer.ahd_homogeneity_cython import build_map as gpu_build_map
    from pysp_amd.dng_warp_corr.dng_warp_rectilinear_coords import compute_remapping_table as gpu_table
    ref_map = load_ref("ahd_homogeneity_cython")
    ref_warp = load_ref("dng_warp_rectilinear_coords")
    H, W = 4000, 6000
    rng = np.random.default_rng(1)
    lab = np.ascontiguousarray(rng.random((H + 2, W + 2, 3), dtype=np.float32) * np.float32(100))
    for name, fn in (("reference build_map (Cython + OpenMP, all host cores)", lambda: ref_map.build_map(lab, 1, 3, False)),
                     ("pysp_build_map_f32 (incl. PCIe)", lambda: gpu_build_map(lab, 1, 3, False))):
        fn(); t0 = time.perf_counter(); out = fn(); dt = time.perf_counter() - t0
        extra = "" if "reference" in name else f", kernel {_lib.default_context().last_kernel_ms():.3f} ms"
        print(f"{name}: {dt * 1e3:.1f} ms = {H * W / 1e6 / dt:.0f} MP/s{extra}")
    a, b = ref_map.build_map(lab, 1, 3, True), gpu_build_map(lab, 1, 3, True)
    print("bit-exact:", bool(np.array_equal(np.asarray(a), b)))
    for name, fn in (("reference compute_remapping_table (Cython + OpenMP)", lambda: ref_warp.compute_remapping_table(1.0, 0.01, 0.002, 0.0, 0.0, 0.0, W, H, 0.5, 0.5, 1.0)),
                     ("pysp_warp_table_f32 (incl. PCIe)", lambda: gpu_table(1.0, 0.01, 0.002, 0.0, 0.0, 0.0, W, H, 0.5, 0.5, 1.0))):
        fn(); t0 = time.perf_counter(); out = fn(); dt = time.perf_counter() - t0
        extra = "" if "reference" in name else f", kernel {_lib.default_context().last_kernel_ms():.3f} ms"
        print(f"{name}: {dt * 1e3:.1f} ms = {H * W / 1e6 / dt:.0f} MP/s{extra}")
    print("cores:", len(os.sched_getaffinity(0)))


if __name__ == "__main__":
    main()
