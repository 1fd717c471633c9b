#!/usr/bin/env python3
"""Build container only: check the product-signed receipts (gpurun_out/product_signed_receipts.json, made on the GPU by
scripts/make_signed_receipt.py) with the REFERENCE's verify_receipt / verify_receipt_mode (core/receipts.py:86-179) and
write receipts + verdicts to tests/golden/product_signed_receipts.json."""
import copy
import json
import os
import sys

sys.path.insert(0, "/root/reference")
from oscillink import verify_receipt  # noqa: E402
from oscillink.core.receipts import verify_receipt_mode  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
src = json.load(open(os.path.join(ROOT, "gpurun_out", "product_signed_receipts.json")))
verdicts = {}
for mode, rec in src["receipts"].items():
    bad = copy.deepcopy(rec)
    bad["meta"]["signature"]["payload"]["deltaH_total"] += 1.0
    verdicts[mode] = {
        "verify_receipt": bool(verify_receipt(rec, src["secret"])),
        "verify_receipt_wrong_secret": bool(verify_receipt(rec, "wrong")),
        "verify_receipt_mode": bool(verify_receipt_mode(rec, src["secret"], require_mode=mode)[0]),
        "tampered": bool(verify_receipt(bad, src["secret"])),
    }
    rec["null_points"] = rec["null_points"][:4]  # keep the fixture small; not part of the signed payload
json.dump({"secret": src["secret"], "receipts": src["receipts"], "reference_verdicts": verdicts},
          open(os.path.join(HERE, "product_signed_receipts.json"), "w"), indent=1, sort_keys=True)
print(verdicts)
