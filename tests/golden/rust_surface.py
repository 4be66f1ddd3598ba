"""A small reader of Rust source for ONE purpose: the PUBLIC surface of a file -- `pub struct` items with their `pub` fields, `pub enum`s with
their variants, `pub trait`s with their method signatures, and `pub fn` items with their owner (`impl` type or module), argument names and
types and return type.  Used twice: tests/golden/make_api_surface.py runs it over the reference's files for the multilinear hot path and
commits the listing (tests/golden/reference_api_surface.json: names and types, each entry citing file:line -- data, no source text), and
tests/test_rust_shim_surface.py runs it over rust_shim/src/lib.rs and compares.  There is no Rust toolchain in the build image, so this listing
is the only mechanical check the shim gets.

Types are compared after normalisation (norm_type): whitespace removed, the generic field / pairing bounds the shim narrows
(`PrimeField` -> `ZkField`, `Pairing` -> `ZkPairing`) are not part of a signature's argument or return types, so they do not appear."""
import re


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def norm_type(t):
    t = re.sub(r"\s+", "", t)
    t = t.replace("&mutself", "&mut self")
    return t


def _match_close(s, i, open_ch, close_ch):
    depth = 0
    for j in range(i, len(s)):
        if s[j] == open_ch:
            depth += 1
        elif s[j] == close_ch:
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced " + open_ch)


def _split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "<([{":
            depth += 1
        elif ch in ">)]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def _fn_sig(src, m_end, name, line, owner):
    """src[m_end] is just behind `fn name`; returns the entry and the index behind the signature"""
    i = m_end
    if src[i] == "<":                                         # generic parameters: not part of the compared surface
        i = _match_close(src, i, "<", ">") + 1
    while src[i].isspace():
        i += 1
    assert src[i] == "(", (name, src[i:i + 20])
    j = _match_close(src, i, "(", ")")
    args = []
    for a in _split_top(src[i + 1:j]):
        a = a.strip()
        if a in ("self", "&self", "&mut self", "mut self"):
            args.append({"name": "self", "type": norm_type(a)})
            continue
        nm, ty = a.split(":", 1)
        args.append({"name": nm.replace("mut ", "").strip(), "type": norm_type(ty)})
    k = j + 1
    rest = src[k:k + 400]
    mret = re.match(r"\s*->\s*", rest)
    ret = ""
    if mret:
        k2 = k + mret.end()
        depth, e = 0, k2
        while e < len(src):
            ch = src[e]
            if ch in "<([":
                depth += 1
            elif ch in ">)]":
                depth -= 1
            elif depth == 0 and (ch in "{;" or src[e:e + 5] == "where"):
                break
            e += 1
        ret = norm_type(src[k2:e])
    return {"kind": "fn", "owner": owner, "name": name, "args": args, "ret": ret, "line": line}


def public_surface(text):
    """-> list of entries: {"kind": "struct"|"enum"|"trait"|"fn", "name", ..., "line"}; nested `mod` names are NOT part of an item's path here
    (the reference spreads its items over crates and files, the shim over modules of one file): items are matched by (owner, name)."""
    src = strip_comments(text)
    line_of = lambda pos: src.count("\n", 0, pos) + 1
    out = []
    # impl blocks and traits: (start, end, owner)
    owners = []
    for m in re.finditer(r"\bimpl\b", src):
        i = m.end()
        while src[i].isspace():
            i += 1
        if src[i] == "<":
            i = _match_close(src, i, "<", ">") + 1
        b = src.index("{", i)
        head = src[i:b]
        head = head.split(" where ")[0].split("\nwhere")[0]
        trait_for = re.match(r"\s*([A-Za-z_][\w:]*)(?:<.*>)?\s+for\s+(.*)", head, re.S)
        target = trait_for.group(2) if trait_for else head
        owner = re.match(r"\s*([A-Za-z_]\w*)", target).group(1)
        owners.append((b, _match_close(src, b, "{", "}"), owner, trait_for.group(1) if trait_for else None))
    for m in re.finditer(r"\bpub\s+trait\s+([A-Za-z_]\w*)", src):
        b = src.index("{", m.end())
        e = _match_close(src, b, "{", "}")
        methods = []
        for f in re.finditer(r"\bfn\s+([A-Za-z_]\w*)", src[b:e]):
            methods.append(_fn_sig(src, b + f.end(), f.group(1), line_of(b + f.start()), m.group(1)))
        out.append({"kind": "trait", "name": m.group(1), "methods": [{k: v for k, v in x.items() if k != "kind"} for x in methods], "line": line_of(m.start())})
        owners.append((b, e, None, "__trait_body__"))
    for m in re.finditer(r"\bpub\s+struct\s+([A-Za-z_]\w*)", src):
        i = m.end()
        if src[i] == "<":
            i = _match_close(src, i, "<", ">") + 1
        rest = src[i:]
        mm = re.match(r"\s*(where[^{;]*)?\s*([{;(])", rest, re.S)
        fields = []
        if mm and mm.group(2) == "{":
            b = i + mm.end() - 1
            e = _match_close(src, b, "{", "}")
            for f in _split_top(src[b + 1:e]):
                f = re.sub(r"#\[[^\]]*\]", "", f).strip()
                fm = re.match(r"pub\s+([A-Za-z_]\w*)\s*:\s*(.*)", f, re.S)
                if fm:
                    fields.append({"name": fm.group(1), "type": norm_type(fm.group(2))})
        out.append({"kind": "struct", "name": m.group(1), "fields": fields, "line": line_of(m.start())})
    for m in re.finditer(r"\bpub\s+enum\s+([A-Za-z_]\w*)", src):
        b = src.index("{", m.end())
        e = _match_close(src, b, "{", "}")
        variants = [re.match(r"([A-Za-z_]\w*)", re.sub(r"#\[[^\]]*\]", "", v).strip()).group(1) for v in _split_top(src[b + 1:e])]
        out.append({"kind": "enum", "name": m.group(1), "variants": variants, "line": line_of(m.start())})
    for m in re.finditer(r"\bpub\s+fn\s+([A-Za-z_]\w*)", src):
        pos = m.start()
        inside = [o for o in owners if o[0] < pos < o[1]]
        if any(o[3] == "__trait_body__" for o in inside):
            continue
        owner = None
        if inside:
            o = max(inside, key=lambda o: o[0])
            owner = o[2]
        out.append(_fn_sig(src, m.end(), m.group(1), line_of(pos), owner))
    # trait impls: `impl Trait for Type { fn ... }` -- methods are public through the trait
    for (b, e, owner, trait) in owners:
        if trait and trait != "__trait_body__":
            for f in re.finditer(r"(?<!pub\s)\bfn\s+([A-Za-z_]\w*)", src[b:e]):
                if re.search(r"pub\s+$", src[max(0, b + f.start() - 8):b + f.start()]):
                    continue
                ent = _fn_sig(src, b + f.end(), f.group(1), line_of(b + f.start()), owner)
                ent["via_trait"] = trait
                out.append(ent)
    # the chain of `pub mod` blocks an item sits in ("" at the top level): the reference's crate / file path, the shim's module path
    mods = []
    for m in re.finditer(r"\bpub\s+mod\s+([A-Za-z_]\w*)\s*\{", src):
        b = m.end() - 1
        mods.append((b, _match_close(src, b, "{", "}"), m.group(1)))
    starts = [0]
    for ln in src.split("\n"):
        starts.append(starts[-1] + len(ln) + 1)
    for it in out:
        pos = starts[it["line"] - 1]
        chain = sorted([mm for mm in mods if mm[0] < pos < mm[1]], key=lambda mm: mm[0])
        it["module"] = "::".join(mm[2] for mm in chain)
    return sorted(out, key=lambda x: x["line"])
