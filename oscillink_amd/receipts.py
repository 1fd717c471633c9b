"""HMAC-SHA256 receipt verification -- host-side half of the reference's oscillink/core/receipts.py:86-179.

A receipt's `meta.signature` block is {"algorithm": "HMAC-SHA256", "payload": {...}, "signature": hex}; the MAC
is taken over the sorted-key JSON of the payload.  Both helpers never raise.
"""
from __future__ import annotations

import hashlib
import hmac
import json
from typing import Optional


def _mac(payload: dict, secret) -> str:
    key = secret.encode("utf-8") if isinstance(secret, str) else secret
    return hmac.new(key, json.dumps(payload, sort_keys=True).encode("utf-8"), hashlib.sha256).hexdigest()


def _block(receipt: dict):
    blk = (receipt.get("meta") or {}).get("signature")
    if not blk or blk.get("algorithm") != "HMAC-SHA256":
        return None
    if blk.get("payload") is None or blk.get("signature") is None:
        return None
    return blk


def verify_receipt(receipt: dict, secret) -> bool:
    """True iff the receipt carries a valid HMAC-SHA256 signature for `secret` (receipts.py:86-113)."""
    try:
        blk = _block(receipt)
        if blk is None:
            return False
        return hmac.compare_digest(_mac(blk["payload"], secret), str(blk["signature"]))
    except Exception:
        return False


def verify_receipt_mode(receipt: dict, secret, require_mode: Optional[str] = None, minimal_subset: bool = False,
                        required_sig_v: Optional[int] = None):
    """(ok, payload) with optional mode / version requirements and minimal-subset fallback (receipts.py:116-179)."""
    try:
        blk = _block(receipt)
        if blk is None:
            return False, None
        payload, claimed = blk["payload"], str(blk["signature"])
        mode = payload.get("mode")
        if require_mode and mode != require_mode:
            return False, None
        if required_sig_v is not None and payload.get("sig_v") != required_sig_v:
            return False, None
        if hmac.compare_digest(_mac(payload, secret), claimed):
            return True, payload
        if minimal_subset and mode == "extended":
            sub = {"sig_v": payload.get("sig_v"), "mode": "minimal", "state_sig": payload.get("state_sig"),
                   "deltaH_total": payload.get("deltaH_total")}
            if hmac.compare_digest(_mac(sub, secret), claimed) and require_mode in (None, "minimal"):
                return True, sub
        return False, None
    except Exception:
        return False, None
