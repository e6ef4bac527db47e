"""gpsiq_kernels_id() of the tree (csrc/Makefile: SHA-256 over DEVSRC), for the profile summaries."""
import hashlib
import os


def kernels_id():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multi-sdr-gps-sim_amd", "csrc")
    devsrc = [l for l in open(os.path.join(csrc, "Makefile")) if l.startswith("DEVSRC")][0].split(":=")[1].split()
    return hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in devsrc)).hexdigest()[:16]
