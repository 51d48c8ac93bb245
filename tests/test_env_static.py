"""CPU: where the library reads the environment (the GPU half is tests/test_env_hostile_gpu.py)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_getenv_only_where_a_context_is_created():
    """grep: outside -DFM_ABLATE blocks the library calls getenv() in fm_ctx_create only."""
    csrc = os.path.join(ROOT, "fast-match_amd", "csrc")
    offenders = []
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        depth_ablate = []
        for no, line in enumerate(open(os.path.join(csrc, f)), 1):
            s = line.strip()
            if s.startswith("#if"):
                depth_ablate.append("FM_ABLATE" in s and not s.startswith("#ifndef"))
            elif s.startswith("#else") and depth_ablate:
                depth_ablate[-1] = False
            elif s.startswith("#endif") and depth_ablate:
                depth_ablate.pop()
            elif "getenv(" in s and not any(depth_ablate) and not s.startswith("//"):
                offenders.append((f, no, s))
    assert all(f == "api_ctx.hip" for f, _, _ in offenders), offenders
    txt = open(os.path.join(csrc, "api_ctx.hip")).read()
    body = txt[txt.index('extern "C" int fm_ctx_create('):]
    body = body[:body.index("\n}\n")]
    assert len(offenders) == body.count("getenv(")            # all of them inside fm_ctx_create
