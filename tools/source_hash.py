"""sha256 (first 16 hex digits) over the kernel sources of the library (nerf_tex_amd/csrc/*.h, *.hip and include/nerftex.h), in name order:
what a committed rocprofv3 summary records (tools/summarize_profile.py) and bench.py compares with the tree it runs from, so that a
quoted `roofline.traffic` that predates a kernel change says so."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16(root: str = ROOT) -> str:
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "nerf_tex_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "nerf_tex_amd", "csrc", "*.hip")))
    for f in files + [os.path.join(root, "include", "nerftex.h")]:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def git_head(root: str = ROOT):
    if os.environ.get("NTX_PROFILE_HEAD"):       # a GPU box has no .git: the build container passes its head along (tools/dev/r5_refresh.sh)
        return os.environ["NTX_PROFILE_HEAD"]
    try:
        import subprocess
        return subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


if __name__ == "__main__":
    print(kernel_sources_sha16(), git_head())
