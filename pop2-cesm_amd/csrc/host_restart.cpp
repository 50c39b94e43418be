// host_restart.cpp -- POP binary restart files (host logic, no HIP): the data format either side of the time step.
//
// Restates the 'bin' path of restart.F90 (write_restart :1095-1715, read_restart :184-1088) and io_binary.F90:
//   <file>       direct-access data, one record = one horizontal slab nx_global x ny_global of r8 in native byte
//                order, no record markers (open_binary :396-399, recl_words = nx_global*ny_global, restart.F90:1262);
//                a 3-D field is km consecutive records (write_real8_3d :2590-2640)
//   <file>.hdr   text: "&GLOBAL" + "name:type:value" lines + "/" (open_binary :414-560), then per field
//                "&SHORT_NAME", long_name / units / grid_loc, "id:int:<first record>", "nfield_dims:int:<2|3>", "/"
//                (define_field_binary :760-880); the separator is ':' (io_binary.F90:59)
// Every rank writes / reads the rows of its own blocks at their place in the global slab (pwrite / pread on the
// shared file), so no gather is needed; rank 0 writes the header.
#include <fcntl.h>
#include <unistd.h>
#include <fstream>
#include <sstream>
#include "pop_internal.hpp"

namespace pop {

// fields of the restart file in the order write_restart defines them (restart.F90:1555-1600); varthick surface layer
std::vector<RestartField> restart_fields(const HostModel &h) {
  std::vector<RestartField> f;
  int rec = 1;
  auto add = [&](const char *name, const char *dev, int tl, int n, int ndims, const char *ln, const char *units, const char *loc, int mask) {
    f.push_back(RestartField{name, dev, tl, n, ndims, rec, ln, units, loc, mask});
    rec += ndims == 3 ? h.km : 1;
  };
  add("UBTROP_CUR", "UBTROP", 1, 0, 2, "U barotropic velocity at current time", "cm/s", "2220", 1);
  add("UBTROP_OLD", "UBTROP", 0, 0, 2, "U barotropic velocity at old time", "cm/s", "2220", 1);
  add("VBTROP_CUR", "VBTROP", 1, 0, 2, "V barotropic velocity at current time", "cm/s", "2220", 1);
  add("VBTROP_OLD", "VBTROP", 0, 0, 2, "V barotropic velocity at old time", "cm/s", "2220", 1);
  add("PSURF_CUR", "PSURF", 1, 0, 2, "surface pressure at current time", "dyne/cm2", "2110", 2);
  add("PSURF_OLD", "PSURF", 0, 0, 2, "surface pressure at old time", "dyne/cm2", "2110", 2);
  add("GRADPX_CUR", "GRADPX", 1, 0, 2, "sfc press gradient in x at current time", "dyne/cm3", "2220", 1);
  add("GRADPX_OLD", "GRADPX", 0, 0, 2, "sfc press gradient in x at old time", "dyne/cm3", "2220", 1);
  add("GRADPY_CUR", "GRADPY", 1, 0, 2, "sfc press gradient in y at current time", "dyne/cm3", "2220", 1);
  add("GRADPY_OLD", "GRADPY", 0, 0, 2, "sfc press gradient in y at old time", "dyne/cm3", "2220", 1);
  add("PGUESS", "PGUESS", 1, 0, 2, "guess for sfc pressure at new time", "dyne/cm2", "2110", 2);
  add("FW_OLD", "FW_OLD", 1, 0, 2, "fresh water input at old time", "", "2110", 2);
  add("FW_FREEZE", "", 1, 0, 2, "water flux due to frazil ice formation", "", "2110", 2);   // no ice formation here: zeros
  add("UVEL_CUR", "UVEL", 1, 0, 3, "U velocity at current time", "cm/s", "3221", 3);
  add("UVEL_OLD", "UVEL", 0, 0, 3, "U velocity at old time", "cm/s", "3221", 3);
  add("VVEL_CUR", "VVEL", 1, 0, 3, "V velocity at current time", "cm/s", "3221", 3);
  add("VVEL_OLD", "VVEL", 0, 0, 3, "V velocity at old time", "cm/s", "3221", 3);
  add("TEMP_CUR", "TRACER", 1, 0, 3, "Potential temperature at current time", "degC", "3111", 4);
  add("SALT_CUR", "TRACER", 1, 1, 3, "Salinity at current time", "msu (g/g)", "3111", 4);
  add("TEMP_OLD", "TRACER", 0, 0, 3, "Potential temperature at old time", "degC", "3111", 4);
  add("SALT_OLD", "TRACER", 0, 1, 3, "Salinity at old time", "msu (g/g)", "3111", 4);
  return f;
}

static std::string list_directed(const std::string &v) { return " " + v; }   // write(line(n:),*) leaves a leading blank

int restart_write_header(const HostModel &h, const std::string &path, const std::vector<RestartAttr> &attrs,
                         const std::vector<RestartField> &fields, std::string &err) {
  std::ofstream o(path + ".hdr");
  if (!o) { err = "cannot open " + path + ".hdr for writing"; return 1; }
  o << "&GLOBAL\n";
  for (const RestartAttr &a : attrs) o << a.name << ':' << a.type << ':' << list_directed(a.value) << '\n';
  o << "/\n";
  for (const RestartField &f : fields) {
    o << '&' << f.name << '\n';
    if (!f.long_name.empty()) o << "long_name:char:" << f.long_name << '\n';
    if (!f.units.empty()) o << "units:char:" << f.units << '\n';
    if (!f.grid_loc.empty()) o << "grid_loc:char:" << f.grid_loc << '\n';
    o << "id:int:" << list_directed(std::to_string(f.id)) << '\n';
    o << "nfield_dims:int:" << list_directed(std::to_string(f.ndims)) << '\n';
    o << "/\n";
  }
  (void)h;
  return o.good() ? 0 : 1;
}

// sections of a header file: name -> (attribute -> value), values trimmed; the type tag is dropped
int restart_parse_header(const std::string &path, std::map<std::string, std::map<std::string, std::string>> &sec, std::string &err) {
  std::ifstream in(path + ".hdr");
  if (!in) { err = "cannot open " + path + ".hdr"; return 1; }
  auto trim = [](std::string s) {
    const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
  };
  std::string line, cur;
  while (std::getline(in, line)) {
    line = trim(line);
    if (line.empty()) continue;
    if (line[0] == '&') { cur = trim(line.substr(1)); sec[cur]; continue; }
    if (line[0] == '/') { cur.clear(); continue; }
    if (cur.empty()) continue;
    const size_t p1 = line.find(':');
    if (p1 == std::string::npos) continue;
    const size_t p2 = line.find(':', p1 + 1);
    if (p2 == std::string::npos) continue;
    sec[cur][trim(line.substr(0, p1))] = trim(line.substr(p2 + 1));
  }
  return 0;
}

static inline double bswap(double v) {
  unsigned char b[8];
  std::memcpy(b, &v, 8);
  for (int i = 0; i < 4; ++i) std::swap(b[i], b[7 - i]);
  std::memcpy(&v, b, 8);
  return v;
}

// one slab (record `rec`, 1-based) <-> the physical cells of the local blocks; `loc` points at the level of local
// block 0 and blocks are `blk_stride` doubles apart (n2 for a 2-D field, n3 for one level of a 3-D field)
int restart_slab_io(const HostModel &h, int fd, long long rec, double *loc, size_t blk_stride, bool write, bool swap, std::string &err) {
  const long long nx = h.c.nx_global, ny = h.c.ny_global, base = (rec - 1) * nx * ny;
  for (int lb = 0; lb < h.nblocks; ++lb) {
    const BlockInfo &B = h.all_blocks[h.local_ids[lb] - 1];
    for (int j = B.jb; j <= B.je; ++j) {
      const int jg = B.j_glob[j - 1];
      if (jg < 1 || jg > ny) continue;
      int i = B.ib;
      while (i <= B.ie) {                      // runs of consecutive global i (a block never wraps inside its physical part)
        const int ig = B.i_glob[i - 1];
        if (ig < 1 || ig > nx) { ++i; continue; }
        int n = 1;
        while (i + n <= B.ie && B.i_glob[i + n - 1] == ig + n) ++n;
        double *p = loc + (size_t)lb * blk_stride + (size_t)(j - 1) * h.nxb + (i - 1);
        const off_t off = (off_t)((base + (long long)(jg - 1) * nx + (ig - 1)) * 8);
        if (write) {
          if (pwrite(fd, p, (size_t)n * 8, off) != (ssize_t)n * 8) { err = "restart: short write"; return 1; }
        } else {
          if (pread(fd, p, (size_t)n * 8, off) != (ssize_t)n * 8) { err = "restart: short read (file smaller than the header says)"; return 1; }
          if (swap) for (int t = 0; t < n; ++t) p[t] = bswap(p[t]);
        }
        i += n;
      }
    }
  }
  return 0;
}

// read_restart :881-935: values outside the ocean are reset (CALCU / CALCT masks in 2-D, k > KMU / KMT in 3-D)
void restart_mask(const HostModel &h, const RestartField &f, double *loc, const std::vector<int> &KMT, const std::vector<int> &KMU) {
  const size_t n2 = h.n2;
  for (int lb = 0; lb < h.nblocks; ++lb)
    for (size_t p = 0; p < n2; ++p) {
      const size_t q = lb * n2 + p;
      if (f.mask == 1 && !(KMU[q] >= 1)) loc[q] = 0.0;
      if (f.mask == 2 && !(KMT[q] >= 1)) loc[q] = 0.0;
      if (f.mask == 3 || f.mask == 4) {
        const int kb = f.mask == 3 ? KMU[q] : KMT[q];
        for (int k = kb + 1; k <= h.km; ++k) loc[(size_t)lb * h.n3 + (size_t)(k - 1) * n2 + p] = 0.0;
      }
    }
}

}  // namespace pop
