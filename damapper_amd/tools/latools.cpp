// latools.cpp -- LAsort / LAcat / LAmerge for damapper's chained .las files.
//
// The reference driver finishes every read block with
//     LAsort [-v] [-a] SORT_PATH/<reads>.<ref>.M@.las ; LAcat [-v] SORT_PATH/<reads>.<ref>.M@.S > <reads>.<ref>.las
// and, for -C,
//     LAsort [-v] [-a] SORT_PATH/<ref>.<reads>.R@.las ; LAmerge [-v] [-a] <ref>.<reads> SORT_PATH/<ref>.<reads>.R@.S.las
// (damapper.c:893-910).  Those programs belong to the DALIGNER package, which is not part of the
// reference tree: PARITY UNPINNED -- no golden vector exists for the sorted file.  What is fixed by
// damapper's own output contract, and implemented here:
//   * a chain (a START/plain record followed by its NEXT records, map.c:2741-2815) moves as a unit
//     and keeps its internal order;
//   * -a ("map order", damapper.c:700-702): units ordered by A-read, then A-position of the unit's
//     first record; without -a ("pile order"): by A-read, B-read, orientation, A-position;
//   * remaining ties: aepos, (bread, comp), bbpos, bepos, diffs, then input order (stable);
//   * header = int64 record count + int32 trace spacing; records are copied byte for byte.
// One binary; the behaviour is selected by the name it is invoked under (LAsort, LAcat, LAmerge).

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <queue>
#include <string>
#include <vector>

namespace {

constexpr int      TRACE_XOVR = 125;     // align.h:21
constexpr uint32_t COMP_FLAG = 0x1, NEXT_FLAG = 0x8;
constexpr size_t   REC = 40;             // bytes of an Overlap on disk (align.c:3098-3099)

const char *prog = "LAsort";

[[noreturn]] void die(const char *fmt, const char *a = "", const char *b = "")
{ fprintf(stderr, "%s: ", prog);
  fprintf(stderr, fmt, a, b);
  fprintf(stderr, "\n");
  exit(1);
}

struct Rec
{ int32_t tlen, diffs, abpos, bbpos, aepos, bepos;
  uint32_t flags;
  int32_t aread, bread, pad;
};
static_assert(sizeof(Rec) == REC, "on-disk record is 40 bytes");

struct Unit                // one chain: bytes [off, off+len) of the file image, nrec records
{ size_t off, len;
  int    nrec;
};

struct LasFile
{ std::string          path;
  std::vector<uint8_t> buf;
  int64_t              novl = 0;
  int32_t              tspace = 0;
  int                  tbytes = 1;
  std::vector<Unit>    units;
};

bool file_exists(const std::string &p)
{ FILE *f = fopen(p.c_str(), "rb");
  if (f == nullptr) return false;
  fclose(f);
  return true;
}

std::string strip_las(const std::string &s)
{ if (s.size() > 4 && s.compare(s.size() - 4, 4, ".las") == 0) return s.substr(0, s.size() - 4);
  return s;
}

// "<x>@<y>" stands for <x>1<y>, <x>2<y>, ... up to the first missing file (the block symbol of
// damapper.c:887-910); "@k" / "@k-l" ranges are not used by damapper and not supported.
std::vector<std::string> expand(const std::string &arg)
{ std::string root = strip_las(arg);
  std::vector<std::string> out;
  size_t at = root.rfind('@');
  if (at == std::string::npos)
    { out.push_back(root);
      return out;
    }
  for (int i = 1;; i++)
    { std::string r = root.substr(0, at) + std::to_string(i) + root.substr(at + 1);
      if (!file_exists(r + ".las")) break;
      out.push_back(r);
    }
  if (out.empty()) die("no files match %s.las", root.c_str());
  return out;
}

void load(LasFile &L, const std::string &root)
{ L.path = root + ".las";
  FILE *f = fopen(L.path.c_str(), "rb");
  if (f == nullptr) die("cannot open %s for reading", L.path.c_str());
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz < 12) die("%s is too short to be a .las file", L.path.c_str());
  L.buf.resize((size_t) sz);
  if (fread(L.buf.data(), 1, (size_t) sz, f) != (size_t) sz) die("read error on %s", L.path.c_str());
  fclose(f);
  memcpy(&L.novl, L.buf.data(), 8);
  memcpy(&L.tspace, L.buf.data() + 8, 4);
  L.tbytes = (L.tspace <= TRACE_XOVR) ? 1 : 2;
  size_t off = 12;
  for (int64_t i = 0; i < L.novl; i++)
    { if (off + REC > L.buf.size()) die("%s is truncated", L.path.c_str());
      Rec r;
      memcpy(&r, L.buf.data() + off, REC);
      if (r.tlen < 0) die("%s is corrupt (negative trace length)", L.path.c_str());
      size_t len = REC + (size_t) r.tlen * L.tbytes;
      if (off + len > L.buf.size()) die("%s is truncated", L.path.c_str());
      if ((r.flags & NEXT_FLAG) && !L.units.empty())
        { L.units.back().len += len;
          L.units.back().nrec += 1;
        }
      else
        L.units.push_back(Unit{off, len, 1});
      off += len;
    }
  if (off != L.buf.size()) die("%s has bytes after its last record", L.path.c_str());
}

struct Key
{ int32_t aread, abpos, aepos, bread, comp, bbpos, bepos, diffs;
};

Key key_of(const LasFile &L, const Unit &u)
{ Rec r;
  memcpy(&r, L.buf.data() + u.off, REC);
  return Key{r.aread, r.abpos, r.aepos, r.bread, (int32_t) (r.flags & COMP_FLAG), r.bbpos, r.bepos, r.diffs};
}

// < 0, 0, > 0
int compare(const Key &x, const Key &y, bool map_order)
{ if (x.aread != y.aread) return x.aread < y.aread ? -1 : 1;
  if (!map_order)
    { if (x.bread != y.bread) return x.bread < y.bread ? -1 : 1;
      if (x.comp != y.comp) return x.comp < y.comp ? -1 : 1;
    }
  if (x.abpos != y.abpos) return x.abpos < y.abpos ? -1 : 1;
  if (x.aepos != y.aepos) return x.aepos < y.aepos ? -1 : 1;
  if (x.bread != y.bread) return x.bread < y.bread ? -1 : 1;
  if (x.comp != y.comp) return x.comp < y.comp ? -1 : 1;
  if (x.bbpos != y.bbpos) return x.bbpos < y.bbpos ? -1 : 1;
  if (x.bepos != y.bepos) return x.bepos < y.bepos ? -1 : 1;
  if (x.diffs != y.diffs) return x.diffs < y.diffs ? -1 : 1;
  return 0;
}

void write_header(FILE *f, int64_t novl, int32_t tspace)
{ fwrite(&novl, 8, 1, f);
  fwrite(&tspace, 4, 1, f);
}

FILE *open_out(const std::string &path)
{ FILE *f = fopen(path.c_str(), "wb");
  if (f == nullptr) die("cannot open %s for writing", path.c_str());
  setvbuf(f, nullptr, _IOFBF, 1 << 20);
  return f;
}

void close_out(FILE *f, const std::string &what)
{ if (fflush(f) != 0 || ferror(f)) die("write error on %s", what.c_str());
  if (f != stdout) fclose(f);
}

int usage_sort()
{ fprintf(stderr, "Usage: LAsort [-va] <align:las> ...\n");
  fprintf(stderr, "      -v: verbose\n");
  fprintf(stderr, "      -a: sort by A-read, A-position (map order); off => A-read, B-read piles\n");
  return 1;
}

int main_sort(int argc, char **argv)
{ bool verbose = false, map_order = false;
  std::vector<std::string> args;
  for (int i = 1; i < argc; i++)
    { if (argv[i][0] == '-' && argv[i][1] != 0)
        { for (const char *c = argv[i] + 1; *c; c++)
            if (*c == 'v') verbose = true;
            else if (*c == 'a') map_order = true;
            else return usage_sort();
        }
      else if (argv[i][0] != 0)
        args.push_back(argv[i]);
    }
  if (args.empty()) return usage_sort();
  for (auto &a : args)
    for (auto &root : expand(a))
      { LasFile L;
        load(L, root);
        std::vector<Key> keys(L.units.size());
        for (size_t i = 0; i < L.units.size(); i++) keys[i] = key_of(L, L.units[i]);
        std::vector<uint32_t> perm(L.units.size());
        for (size_t i = 0; i < perm.size(); i++) perm[i] = (uint32_t) i;
        std::stable_sort(perm.begin(), perm.end(),
                         [&](uint32_t x, uint32_t y) { return compare(keys[x], keys[y], map_order) < 0; });
        std::string out = root + ".S.las";
        FILE *f = open_out(out);
        write_header(f, L.novl, L.tspace);
        for (uint32_t p : perm) fwrite(L.buf.data() + L.units[p].off, 1, L.units[p].len, f);
        close_out(f, out);
        if (verbose)
          { printf("  %s: %lld records in %zu chains sorted\n", L.path.c_str(), (long long) L.novl, L.units.size());
            fflush(stdout);
          }
      }
  return 0;
}

int usage_cat()
{ fprintf(stderr, "Usage: LAcat [-v] <source:las> ... > <target>.las\n");
  return 1;
}

int main_cat(int argc, char **argv)
{ bool verbose = false;
  std::vector<std::string> roots;
  for (int i = 1; i < argc; i++)
    { if (argv[i][0] == '-' && argv[i][1] != 0)
        { for (const char *c = argv[i] + 1; *c; c++)
            if (*c == 'v') verbose = true;
            else return usage_cat();
        }
      else if (argv[i][0] != 0)
        for (auto &r : expand(argv[i])) roots.push_back(r);
    }
  if (roots.empty()) return usage_cat();
  // first pass: headers only
  int64_t total = 0;
  int32_t tspace = 0;
  for (size_t i = 0; i < roots.size(); i++)
    { std::string p = roots[i] + ".las";
      FILE *f = fopen(p.c_str(), "rb");
      if (f == nullptr) die("cannot open %s for reading", p.c_str());
      int64_t n;
      int32_t t;
      if (fread(&n, 8, 1, f) != 1 || fread(&t, 4, 1, f) != 1) die("%s is too short to be a .las file", p.c_str());
      fclose(f);
      if (i > 0 && t != tspace) die("%s has a different trace spacing than the files before it", p.c_str());
      tspace = t;
      total += n;
    }
  write_header(stdout, total, tspace);
  std::vector<uint8_t> buf(1 << 22);
  for (auto &r : roots)
    { std::string p = r + ".las";
      FILE *f = fopen(p.c_str(), "rb");
      if (f == nullptr) die("cannot open %s for reading", p.c_str());
      fseek(f, 12, SEEK_SET);
      size_t n;
      while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) fwrite(buf.data(), 1, n, stdout);
      fclose(f);
      if (verbose) fprintf(stderr, "  %s appended\n", p.c_str());
    }
  close_out(stdout, "<stdout>");
  return 0;
}

int usage_merge()
{ fprintf(stderr, "Usage: LAmerge [-va] <merge:las> <parts:las> ...\n");
  return 1;
}

int main_merge(int argc, char **argv)
{ bool verbose = false, map_order = false;
  std::vector<std::string> args;
  for (int i = 1; i < argc; i++)
    { if (argv[i][0] == '-' && argv[i][1] != 0)
        { for (const char *c = argv[i] + 1; *c; c++)
            if (*c == 'v') verbose = true;
            else if (*c == 'a') map_order = true;
            else return usage_merge();
        }
      else if (argv[i][0] != 0)
        args.push_back(argv[i]);
    }
  if (args.size() < 2) return usage_merge();
  std::string target = strip_las(args[0]) + ".las";
  std::vector<LasFile> parts;
  for (size_t i = 1; i < args.size(); i++)
    for (auto &r : expand(args[i]))
      { parts.emplace_back();
        load(parts.back(), r);
      }
  int64_t total = 0;
  int32_t tspace = parts[0].tspace;
  for (auto &p : parts)
    { if (p.tspace != tspace) die("%s has a different trace spacing than the files before it", p.path.c_str());
      total += p.novl;
    }
  struct Head { Key k; int file; size_t unit; };
  auto later = [&](const Head &x, const Head &y)
    { int c = compare(x.k, y.k, map_order);
      if (c != 0) return c > 0;
      return x.file > y.file;                    // equal keys: the earlier part first
    };
  std::priority_queue<Head, std::vector<Head>, decltype(later)> heap(later);
  for (size_t i = 0; i < parts.size(); i++)
    if (!parts[i].units.empty()) heap.push(Head{key_of(parts[i], parts[i].units[0]), (int) i, 0});
  FILE *f = open_out(target);
  write_header(f, total, tspace);
  while (!heap.empty())
    { Head h = heap.top();
      heap.pop();
      LasFile &L = parts[h.file];
      fwrite(L.buf.data() + L.units[h.unit].off, 1, L.units[h.unit].len, f);
      if (h.unit + 1 < L.units.size())
        heap.push(Head{key_of(L, L.units[h.unit + 1]), h.file, h.unit + 1});
    }
  close_out(f, target);
  if (verbose)
    { printf("  %s: %lld records merged from %zu files\n", target.c_str(), (long long) total, parts.size());
      fflush(stdout);
    }
  return 0;
}

}  // namespace

int main(int argc, char **argv)
{ const char *base = strrchr(argv[0], '/');
  base = base ? base + 1 : argv[0];
  prog = base;
  if (strcmp(base, "LAcat") == 0) return main_cat(argc, argv);
  if (strcmp(base, "LAmerge") == 0) return main_merge(argc, argv);
  if (strcmp(base, "LAsort") == 0) return main_sort(argc, argv);
  // generic entry: latools <sort|cat|merge> args...
  if (argc >= 2)
    { std::string cmd = argv[1];
      if (cmd == "sort")  { prog = "LAsort";  return main_sort(argc - 1, argv + 1); }
      if (cmd == "cat")   { prog = "LAcat";   return main_cat(argc - 1, argv + 1); }
      if (cmd == "merge") { prog = "LAmerge"; return main_merge(argc - 1, argv + 1); }
    }
  fprintf(stderr, "Usage: latools <sort|cat|merge> ...   (or invoke as LAsort / LAcat / LAmerge)\n");
  return 1;
}
