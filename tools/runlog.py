"""First line of every measuring tool's output: what ran.  A log that does not say which command, which library and which
tuning switches produced it cannot be attributed afterwards (round 2 lost a GPU fault's cause that way).
Usage: `from runlog import banner; banner()` right after the imports."""
import os
import sys
import time

KEYS = ("VARIANTS", "ROUNDS", "REPS", "BATCH", "DTYPE", "SHAPES", "EPI_STORE", "PAD", "YARDSTICK", "ABLATION_LIB", "VARIANT")


def banner(extra: str = "") -> None:
    env = {k: v for k, v in os.environ.items() if k.startswith(("HM_", "HAMER_", "HIP_", "HSA_", "ROCR_")) or k in KEYS}
    lib = ""
    try:
        from hamer_yolo_amd import lib as L
        lib = f" lib={os.path.basename(L.LIB_PATH)}"
        if os.path.exists(L.LIB_PATH):
            lib += f"@{time.strftime('%H:%M:%S', time.gmtime(os.path.getmtime(L.LIB_PATH)))}"
    except Exception:
        pass
    print(f"# {time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime())} argv={sys.argv!r} env={env!r}{lib} {extra}".rstrip(), flush=True)
