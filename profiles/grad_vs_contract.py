"""Every gradient comparison of a VCR_GRAD_REPORT log (tests/util.py::_report) against BASELINE.json's figure: max-norm relative
error < 1e-4, on ALL rows and on the rows the oracle does not mark fragile.  Lists the comparisons that miss it, worst first,
with the tensor (= the stage whose adjoint produces it) and the fraction of rows the fragile mask removed.
    python profiles/grad_vs_contract.py <report> > profiles/r6_grad_vs_1e-4.txt"""
import collections
import re
import sys

CONTRACT = 1e-4
STAGE = {"means3D": "projection bwd (mean: Jacobian + view direction)", "xyz": "projection bwd (mean)", "scales": "projection bwd (Sigma3D -> scale)",
         "scaling": "projection bwd + activation", "rots": "projection bwd (Sigma3D -> quaternion)", "rotation": "projection bwd + activation",
         "opac": "compositing bwd (sum p) / opacity", "opacity": "compositing bwd + sigmoid", "shs": "compositing bwd (sum w g) x SH basis",
         "f_dc": "SH basis", "f_rest": "SH basis", "normals": "compositing bwd (plane / normal adjoint)", "m2": "compositing bwd (sum u p)",
         "m2d": "compositing bwd (sum |u p|)", "sem": "compositing bwd (semantics)", "obj_dc": "semantics"}
def main():
    rows = []
    for line in open(sys.argv[1]):
        cols = [c.strip() for c in line.split("|")]
        if len(cols) < 4:
            continue
        m = re.match(r"(.*) \[(all|non-fragile (\d+)/(\d+))\]", cols[1])
        mx = re.search(r"maxnorm ([0-9.e+-]+)/", line)
        if not m or not mx:
            continue
        name, part = m.group(1), ("all" if m.group(2) == "all" else "non-fragile")
        kept = (int(m.group(3)), int(m.group(4))) if m.group(3) else None
        rows.append(dict(test=cols[0].split("::")[-1], name=name, part=part, kept=kept, maxnorm=float(mx.group(1))))
    by = collections.defaultdict(dict)
    for r in rows:
        by[(r["test"], r["name"])][r["part"]] = r
    n_all = n_nf = miss_all = miss_nf = 0
    out = []
    for (test, name), parts in by.items():
        a, nf = parts.get("all"), parts.get("non-fragile")
        if a:
            n_all += 1
            miss_all += a["maxnorm"] >= CONTRACT
        if nf:
            n_nf += 1
            miss_nf += nf["maxnorm"] >= CONTRACT
        worst = max(p["maxnorm"] for p in parts.values())
        if worst >= CONTRACT:
            key = name.split(":")[-1]
            frag = "" if not nf else f"{1 - nf['kept'][0] / max(nf['kept'][1], 1):.1%} fragile"
            out.append((worst, f"{(nf['maxnorm'] if nf else float('nan')):.2e}", f"{(a['maxnorm'] if a else float('nan')):.2e}", frag, name, test,
                        STAGE.get(key, "-")))
    print(f"# {sys.argv[1]}: {len(by)} tensor comparisons; contract = max-norm relative error < {CONTRACT:g} against the fp64 oracle")
    print(f"# on ALL rows: {n_all - miss_all} of {n_all} meet it; on the NON-FRAGILE rows (where a mask exists): {n_nf - miss_nf} of {n_nf} meet it")
    print("# comparisons that miss it on either set, worst first:  non-fragile | all | fragile share | tensor | test | stage of the adjoint")
    for w, a, b, f, name, test, stage in sorted(out, reverse=True):
        print(f"{a:>9s} | {b:>9s} | {f:>14s} | {name} | {test} | {stage}")


if __name__ == "__main__":
    main()
