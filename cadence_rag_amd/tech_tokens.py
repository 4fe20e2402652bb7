"""Technical-token extraction of the exact-match lane — the regexes and the domain lexicon are the
reference's data (/root/reference/app/ingest.py:24-73), the procedure follows extract_tech_tokens
(ingest.py:141-160): all pattern matches in pattern order, then one canonical token per matching
lexicon rule, stripped, de-duplicated case-insensitively keeping first occurrences.  Pinned by
tests/golden/reference_host_logic.json.  ("next" row of SURVEY.md 8f: hybrid fusion, config 5.)"""
from __future__ import annotations

import re
from typing import List

_I = re.IGNORECASE
TECH_TOKEN_PATTERNS = [
    re.compile(r"https?://\S+", _I),
    re.compile(r"\b(?:\d{1,3}\.){3}\d{1,3}\b"),
    re.compile(r"\b[A-Z]{2,10}-\d+\b"),
    re.compile(r"\bE[A-Z0-9_]{2,}\b"),
    re.compile(r"\bHTTP\s?\d{3}\b", _I),
    re.compile(r"\bORA-\d{4,}\b", _I),
    re.compile(r"\bv?\d+\.\d+(?:\.\d+)?\b"),
    re.compile(r"\b[a-f0-9]{7,40}\b", _I),
    re.compile(r"(?:/[\w.\-]+)+"),
]

DOMAIN_TECH_TOKEN_RULES = [(re.compile(rx, _I), canon) for rx, canon in [
    (r"\bbill of materials\b", "BOM"), (r"\bbom\b", "BOM"), (r"\bbuild(?:s|ing)?\b", "build"),
    (r"\bssd\b", "SSD"), (r"\bobject\s+(?:store|storage)\b", "object store"), (r"\bobject\b", "object"),
    (r"\btiering\b", "tiering"), (r"\blenovo\b", "Lenovo"), (r"\bdell\b", "Dell"),
    (r"\bsuper[\s-]?micro\b|\bsmc\b", "Supermicro"), (r"\baws\b|\bamazon web services\b", "AWS"),
    (r"\bamazon\b", "Amazon"), (r"\bazure\b", "Azure"), (r"\bmicrosoft\b", "Microsoft"),
    (r"\bgcp\b|\bgoogle cloud(?: platform)?\b", "GCP"), (r"\bgoogle\b", "Google"),
    (r"\boci\b|\boracle cloud(?: infrastructure)?\b", "OCI"), (r"\boracle\b", "Oracle"),
    (r"\bcompet(?:e|es|ing|ition|itive|itor|itors)\b", "competitive"), (r"\bincumbent\b", "incumbent"),
    (r"\bbake[\s-]?off\b", "bake-off"), (r"\bhead[\s-]?to[\s-]?head\b", "head-to-head"),
    (r"\bvs\.?(?=\s|$)|\bversus\b", "vs"),
]]


def extract_tech_tokens(text: str) -> List[str]:
    found: List[str] = []
    for pat in TECH_TOKEN_PATTERNS:
        found.extend(pat.findall(text))
    found.extend(canon for pat, canon in DOMAIN_TECH_TOKEN_RULES if pat.search(text))
    seen, out = set(), []
    for tok in (t.strip() for t in found):
        if tok and tok.lower() not in seen:
            seen.add(tok.lower())
            out.append(tok)
    return out
