"""The reference's main at full size through both hosts: the C++ command-line program and the Python
render() must write the same 1200x800 spp-500 PPM, byte for byte (both take the text from the device)."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracinginoneweekend_amd import raytracer as R
from raytracinginoneweekend_amd.basetypes import Image
exe = os.path.join(ROOT, "raytracinginoneweekend_amd", "lib", "rtow_raytracer")
t0 = time.time()
subprocess.check_call([exe, "--width", "1200", "--ratio", "1.5", "--spp", "500", "--depth", "50", "--out", "/tmp/cli.ppm"])
t1 = time.time()
R.render(Image(1200, 800), "/tmp/py.ppm", R.randomScene(), R.bookCamera(1.5), 500, 50)
t2 = time.time()
a, b = open("/tmp/cli.ppm", "rb").read(), open("/tmp/py.ppm", "rb").read()
print(f"C++ host {t1 - t0:.2f} s, Python host {t2 - t1:.2f} s, {len(a)} bytes, sha1 {hashlib.sha1(a).hexdigest()}, equal {a == b}")
print(a[:40])
