import subprocess, sys, torch
print("cuda", torch.cuda.is_available(), flush=True)
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
r = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True, timeout=60)
print("rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:], flush=True)
