"""include/hnsw_mi355x.h is the ONE description of the C ABI: this module reads it and derives the ctypes side
(structures and prototypes), so that the Python binding cannot drift from the header a C / Rust / Julia host compiles
against.  Only what the header uses is understood: typedef'd plain structs, opaque struct typedefs, one function-pointer
typedef, prototypes with scalar / pointer parameters.
"""
import ctypes as C
import re

SCALARS = {
    "int": C.c_int, "unsigned": C.c_uint, "unsigned int": C.c_uint, "float": C.c_float, "double": C.c_double,
    "char": C.c_char, "size_t": C.c_size_t, "int8_t": C.c_int8, "uint8_t": C.c_uint8, "int32_t": C.c_int32,
    "uint32_t": C.c_uint32, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _norm(t):
    """'const  float *' -> ('float', 1): base type without qualifiers, pointer depth."""
    depth = t.count("*")
    base = " ".join(w for w in t.replace("*", " ").split() if w not in ("const", "struct", "volatile"))
    return base, depth


class Header:
    """structs: name -> ctypes.Structure subclass (complete types only); opaque: names of opaque struct typedefs;
    prototypes: name -> (restype, [argtypes], [arg names], 'C text')."""

    def __init__(self, text):
        self.text = strip_comments(text)
        self.structs = {}
        self.opaque = set()
        self.fnptr = set()
        self.prototypes = {}
        self.constants = {}
        self._parse_typedefs()
        self._parse_prototypes()

    # ---- types
    def ctype(self, decl, for_return=False):
        base, depth = _norm(decl)
        if depth == 0:
            if base == "void":
                return None
            if base in SCALARS:
                return SCALARS[base]
            if base in self.structs:
                return self.structs[base]
            if base in self.fnptr:
                return C.c_void_p
            raise ValueError(f"hnsw_mi355x.h: type '{decl}' is not understood by the binding generator")
        if depth == 1 and base == "char":
            return C.c_char_p
        if depth == 1 and base in self.structs:
            return C.POINTER(self.structs[base])
        if base not in SCALARS and base not in self.opaque and base not in self.structs and base != "void":
            raise ValueError(f"hnsw_mi355x.h: pointer to unknown type '{decl}'")
        return C.c_void_p  # buffers, opaque handles, out-pointers: plain addresses (byref() and c_void_p both fit)

    def _parse_typedefs(self):
        t = self.text
        for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", t):
            self.opaque.add(m.group(2))
        for m in re.finditer(r"typedef\s+\w[\w\s\*]*\(\s*\*\s*(\w+)\s*\)\s*\([^)]*\)\s*;", t):
            self.fnptr.add(m.group(1))
        for m in re.finditer(r"typedef\s+struct\s*\w*\s*\{([^{}]*)\}\s*(\w+)\s*;", t):
            body, name = m.group(1), m.group(2)
            fields = []
            for decl in body.split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                am = re.match(r"(.+?)\s*(\w+)\s*\[\s*(\d+)\s*\]$", decl)
                if am:
                    fields.append((am.group(2), self.ctype(am.group(1)) * int(am.group(3))))
                    continue
                fm = re.match(r"(.+?[\s\*])(\w+)$", decl)
                if not fm:
                    raise ValueError(f"hnsw_mi355x.h: struct {name}: cannot read field '{decl}'")
                fields.append((fm.group(2), self.ctype(fm.group(1))))
            self.structs[name] = type(name, (C.Structure,), {"_fields_": fields})
        for m in re.finditer(r"#define\s+(HNSWGPU_\w+)\s+(-?\d+)\b", t):
            self.constants[m.group(1)] = int(m.group(2))
        for m in re.finditer(r"\b(HNSWGPU_\w+)\s*=\s*(-?\d+)", t):  # enum members written with their values
            self.constants.setdefault(m.group(1), int(m.group(2)))

    def _parse_prototypes(self):
        pat = re.compile(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][\w ]*?[\s\*]+)([A-Za-z_]\w*)\s*\(([^;{}()]*)\)\s*;")
        for m in pat.finditer(self.text):
            ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
            if ret.split()[0] in ("typedef", "return"):
                continue
            argtypes, names = [], []
            if args and args != "void":
                for a in args.split(","):
                    a = a.strip()
                    am = re.match(r"(.+?[\s\*])(\w+)$", a)
                    if not am:
                        raise ValueError(f"hnsw_mi355x.h: {name}: cannot read parameter '{a}'")
                    argtypes.append(self.ctype(am.group(1)))
                    names.append(am.group(2))
            self.prototypes[name] = (self.ctype(ret, for_return=True), argtypes, names, f"{' '.join(ret.split())} {name}({args})")


def load(path):
    with open(path) as f:
        return Header(f.read())
