#!/usr/bin/env python
"""Scan a gfx950 .s file (hipcc -S --cuda-device-only) for wait-state hazards the compiler should have padded:
  * VALU write of vX  ->  v_permlane16/32_swap reading vX           : 2 wait states   (cdna_hip_programming.md T21)
  * VALU write of vX  ->  v_readfirstlane / v_readlane reading vX    : 1 wait state    (§5.7 item 2)
  * trans (v_exp/v_rcp/v_rsq/v_log/v_sqrt/v_sin/v_cos) result -> next VALU reading it : 1 wait state (VALUTransUseHazard)
Prints every site where fewer independent issue states separate producer and consumer.  Used in round 5 to look for the cause of the
run-to-run differences of the fused dQ epilogue (DESIGN / profiles/r05_nondeterminism.md)."""
import re, sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None: out.add(int(m.group(1)))
        else: out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out
TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_log_", "v_sqrt_", "v_sin_", "v_cos_")

def main(path):
    kern = None
    hist = []   # (mnemonic, written vregs, states)  most recent last
    n_sites = 0
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].strip()
        if not s: continue
        if s.endswith(":") and not s.startswith("."):
            if s.startswith("_Z") or s.startswith("qfx") : kern = s[:-1]; hist = []
            continue
        if s.startswith("."): continue
        parts = s.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if mn == "s_nop":
            hist.append(("s_nop", set(), int(ops[0], 0) + 1)); continue
        is_valu = mn.startswith("v_") and not mn.startswith("v_mfma") and not mn.startswith("v_smfma")
        reads = set()
        writes = set()
        if mn.startswith("v_permlane") and "swap" in mn:
            reads = regs(ops[0]) | regs(ops[1]); writes = set(reads); need = 2
        elif mn.startswith("v_readfirstlane") or mn.startswith("v_readlane"):
            reads = regs(ops[1]); need = 1
        elif is_valu:
            writes = regs(ops[0]) if ops else set()
            for o in ops[1:]: reads |= regs(o)
            need = 0
        else:
            need = 0
        # check
        if reads:
            dist = 0
            for (pm, pw, st) in reversed(hist[-6:]):
                if pw & reads:
                    req = need if not pm.startswith(TRANS) else max(need, 1 if is_valu else 0)
                    if (pm.startswith("v_") and not pm.startswith("v_mfma")) and dist < req:
                        print(f"{path}:{ln}: [{kern}] {pm} -> {mn} on v{sorted(pw & reads)} with {dist} states (need {req})")
                        n_sites += 1
                    break
                dist += st
        hist.append((mn, writes if is_valu else set(), 1))
        if len(hist) > 16: hist = hist[-16:]
    print(f"{path}: {n_sites} suspicious sites")

if __name__ == "__main__":
    for p in sys.argv[1:]: main(p)
