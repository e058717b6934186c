"""GPU diagnostic: find the smallest set of preceding tests that makes a target test fail (order-dependent failure)."""
import subprocess, sys, os
target = sys.argv[1]
out = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q"], capture_output=True, text=True).stdout
ids = [l.strip() for l in out.splitlines() if "::" in l]
ti = next(i for i, x in enumerate(ids) if target in x)
pre = ids[:ti]
def fails(subset):
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider"] + subset + [ids[ti]], capture_output=True, text=True)
    return "failed" in r.stdout.splitlines()[-1]
print("full prefix fails:", fails(pre), flush=True)
cur = pre
while len(cur) > 1:
    h = len(cur) // 2
    a, b = cur[:h], cur[h:]
    if fails(b): cur = b
    elif fails(a): cur = a
    else:
        print("needs tests from both halves; stopping at", len(cur)); break
    print("narrowed to", len(cur), flush=True)
print("\n".join(cur))
