"""A/B timing of f16c6 builds in ONE process per library, alternating: ab_c6.py libA libB [rounds] (use 'default' for the in-tree build)."""
import subprocess, sys, os
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["GENOMAD_AMD_LIB"] = lib
        else:
            env.pop("GENOMAD_AMD_LIB", None)
        out = subprocess.run([sys.executable, "scripts/ablate_c6.py", "16384"], env=env, capture_output=True, text=True).stdout.strip()
        print(out, flush=True)
