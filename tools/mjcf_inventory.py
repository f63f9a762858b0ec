#!/usr/bin/env python
"""Distance between the MJCF importer and the reference's task files: `mjcf.dry_run` over every XML under
<reference>/myosuite/envs/myo/assets (includes into the empty simhive/myo_sim submodule are recorded, not followed).

    python tools/mjcf_inventory.py [/root/reference] > profiles/r05_mjcf_dry_run.json

Runs where the reference checkout is present (this container); the committed JSON is what travels."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myosuite_amd.model import mjcf   # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    base = os.path.join(ref, "myosuite", "envs", "myo", "assets")
    files = sorted(os.path.join(dp, f) for dp, _, fs in os.walk(base) for f in fs if f.endswith(".xml"))
    out = {"_what": "mjcf.dry_run over the reference's task XMLs; `unsupported` = constructs mjcf.load rejects (physics the engine "
                    "does not implement), `ignored` = non-physical elements it skips, `missing_includes` = files of the empty "
                    "simhive submodules the task file points at", "files": {}}
    tot = {}
    for f in files:
        rel = os.path.relpath(f, base)
        try:
            r = mjcf.dry_run(f)
            r["file"] = rel
            r["unsupported"] = {k: (v if len(v) <= 6 else v[:6] + [f"... {len(v) - 6} more"]) for k, v in r["unsupported"].items()}
            for k, v in r["unsupported"].items():
                tot[k] = tot.get(k, 0) + 1
            # and the importer itself on what IS in the checkout (includes skipped, lenient about what only they define)
            try:
                sp = mjcf.load(f, missing_include="skip")
                r["load_with_includes_skipped"] = {"ok": True, "bodies": len(sp.bodies), "joints": len(sp.joints), "skipped": getattr(sp, "import_skipped", {})}
            except mjcf.MjcfError as exc:
                r["load_with_includes_skipped"] = {"ok": False, "error": str(exc)[:200]}
        except Exception as exc:   # a file the walker itself cannot read is a finding, not a crash
            r = {"file": rel, "error": repr(exc)}
        out["files"][rel] = r
    out["unsupported_constructs_by_number_of_files"] = dict(sorted(tot.items(), key=lambda kv: -kv[1]))
    out["importer_rejections_with_includes_skipped"] = {k: r["load_with_includes_skipped"]["error"] for k, r in out["files"].items()
                                                         if "load_with_includes_skipped" in r and not r["load_with_includes_skipped"]["ok"]}
    out["files_total"] = len(files)
    out["files_with_nothing_unsupported_in_the_task_file_itself"] = sum(1 for r in out["files"].values() if not r.get("unsupported") and "error" not in r)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
