"""Do the counter summaries under profiles/ belong to the kernel sources in the tree?  One line per summary:
    python tools/check_profiles.py [round]        (default r02)
bench.py uses a summary only when its `source_sha` matches (tools/kernel_sha.py); after editing a kernel family's
sources, re-collect that family (`bash tools/pmc_collect.sh <round> <workload>` on the GPU box, then
`bash tools/stash_profiles.sh <round> <workload>`)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_sha import kernel_source_sha  # noqa: E402


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    stale = 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}_pmc_*.json"))):
        d = json.load(open(f))
        name = d.get("workload", "")
        sha = kernel_source_sha(ROOT, name)
        ok = d.get("source_sha") == sha
        stale += not ok
        print(f"{'ok   ' if ok else 'STALE'} {os.path.basename(f)}  {d.get('dominant_kernel')}  summary {d.get('source_sha')} tree {sha}")
    return 1 if stale else 0


if __name__ == "__main__":
    sys.exit(main())
