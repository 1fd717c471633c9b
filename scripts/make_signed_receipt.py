#!/usr/bin/env python3
"""Runs on the MI355X box: sign receipts (minimal + extended payloads) with the product, write them to
gpurun_out/product_signed_receipts.json.  tests/golden/verify_signed_with_reference.py then checks them with the
reference's verifier in the build container and commits the verdicts as a fixture."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oscillink_amd import Oscillink  # noqa: E402
from tests._cases import load_case, make_inputs  # noqa: E402

SECRET = "oscillink-golden-secret"
out = {"secret": SECRET, "receipts": {}}
case = load_case("g1_n400_d64_k6_chain8")
rc = case["recipe"]
Y, psi = make_inputs(rc)
for mode in ("minimal", "extended"):
    lat = Oscillink(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"])
    lat.set_query(psi)
    lat.add_chain(rc["chain"], lamP=rc["lamP"])
    lat.set_receipt_secret(SECRET)
    lat.set_signature_mode(mode)
    lat.settle(max_iters=rc["settle_max_iters"], tol=rc["settle_tol"])
    out["receipts"][mode] = lat.receipt()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "product_signed_receipts.json"), "w"))
print("wrote", {m: r["meta"]["signature"]["signature"][:16] for m, r in out["receipts"].items()})
