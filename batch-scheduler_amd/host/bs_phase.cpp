// bs_phase.cpp — the PodGroup phase machine (pkg/scheduler/controller/controller.go:179-311, with the in-memory transitions of core.go:279-281,
// :325-360 and the gate of batchscheduler.go:258-285) and the JSON merge-patch writer status changes travel in (pkg/util/k8s.go:34-48 over
// evanphx/json-patch v4.5.0+incompatible, go.mod:33).  Host-only: no GPU, no HIP.  Interface and the reference lines behind every entry point:
// include/bsched_host.h.  SURVEY.md section 8(f)-4, second half; pinned on pkg/util/k8s_test.go:31-78 (tests/test_phase_machine.py).
//
// Recalled upstream behaviour (neither module is vendored in the reference tree; listed here so that a reader can check it against the sources):
//   P1  jsonpatch.CreateMergePatch unmarshals both texts into map[string]interface{} and calls getDiff(a, b): for every key of b — absent in a:
//       taken; other dynamic type: taken; object: recurse, taken when the sub-diff is non-empty; string / float64 / bool: taken when different;
//       array: taken whole unless matchesArray (same length, element-wise deep equality); null: taken unless a's is null too — then every key
//       only a has becomes null.  The result is json.Marshal'ed.
//   P2  encoding/json writes map keys sorted by byte order, no white space, float64 through strconv 'f' with the shortest round-trip digits
//       ('e' below 1e-6 and from 1e21 on, a two-digit negative exponent's leading zero dropped), strings with \" \\ \n \r \t, other control
//       characters as \u00XX, and <, >, &, U+2028, U+2029 escaped (HTML-safe by default).
//   P3  metav1.Time marshals as null when IsZero(), else as UTC RFC 3339 with second precision.
#include "../../include/bsched.h"
#include "../../include/bsched_host.h"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------ JSON values
struct J {
  enum Kind : uint8_t { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double n = 0;
  std::string s;
  std::vector<J> a;
  std::map<std::string, J> o;      // (std::map iterates in byte order of the keys: encoding/json's order for maps)
};

struct Parser {
  const char* p;
  const char* end;
  bool ok = true;
  void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
  bool lit(const char* w) {
    const size_t n = std::strlen(w);
    if ((size_t)(end - p) >= n && std::memcmp(p, w, n) == 0) { p += n; return true; }
    return false;
  }
  static void utf8(std::string& out, uint32_t c) {
    if (c < 0x80) out += (char)c;
    else if (c < 0x800) { out += (char)(0xC0 | (c >> 6)); out += (char)(0x80 | (c & 0x3F)); }
    else if (c < 0x10000) { out += (char)(0xE0 | (c >> 12)); out += (char)(0x80 | ((c >> 6) & 0x3F)); out += (char)(0x80 | (c & 0x3F)); }
    else { out += (char)(0xF0 | (c >> 18)); out += (char)(0x80 | ((c >> 12) & 0x3F)); out += (char)(0x80 | ((c >> 6) & 0x3F)); out += (char)(0x80 | (c & 0x3F)); }
  }
  bool hex4(uint32_t& v) {
    if (end - p < 4) return false;
    v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = p[i];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
      else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
      else return false;
    }
    p += 4;
    return true;
  }
  bool str(std::string& out) {
    if (p >= end || *p != '"') return false;
    ++p;
    while (p < end && *p != '"') {
      if ((unsigned char)*p < 0x20) return false;
      if (*p != '\\') { out += *p++; continue; }
      if (++p >= end) return false;
      switch (*p++) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          uint32_t c, d;
          if (!hex4(c)) return false;
          if (c >= 0xD800 && c < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {        // surrogate pair
            const char* save = p;
            p += 2;
            if (hex4(d) && d >= 0xDC00 && d < 0xE000) c = 0x10000 + ((c - 0xD800) << 10) + (d - 0xDC00);
            else { p = save; c = 0xFFFD; }
          } else if (c >= 0xD800 && c < 0xE000) c = 0xFFFD;
          utf8(out, c);
          break;
        }
        default: return false;
      }
    }
    if (p >= end) return false;
    ++p;
    return true;
  }
  bool value(J& v, int depth = 0) {
    if (depth > 256) return false;
    ws();
    if (p >= end) return false;
    if (*p == '{') {
      ++p;
      v.kind = J::Obj;
      ws();
      if (p < end && *p == '}') { ++p; return true; }
      for (;;) {
        ws();
        std::string k;
        if (!str(k)) return false;
        ws();
        if (p >= end || *p++ != ':') return false;
        J child;
        if (!value(child, depth + 1)) return false;
        v.o[k] = std::move(child);          // (a repeated key: the last one wins, as in Go)
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == '}') { ++p; return true; }
        return false;
      }
    }
    if (*p == '[') {
      ++p;
      v.kind = J::Arr;
      ws();
      if (p < end && *p == ']') { ++p; return true; }
      for (;;) {
        J child;
        if (!value(child, depth + 1)) return false;
        v.a.push_back(std::move(child));
        ws();
        if (p < end && *p == ',') { ++p; continue; }
        if (p < end && *p == ']') { ++p; return true; }
        return false;
      }
    }
    if (*p == '"') { v.kind = J::Str; return str(v.s); }
    if (lit("true")) { v.kind = J::Bool; v.b = true; return true; }
    if (lit("false")) { v.kind = J::Bool; v.b = false; return true; }
    if (lit("null")) { v.kind = J::Null; return true; }
    // number (RFC 8259 grammar; the value is the nearest float64, as interface{} decoding gives)
    const char* s = p;
    if (p < end && *p == '-') ++p;
    if (p >= end) return false;
    if (*p == '0') ++p;
    else if (*p >= '1' && *p <= '9') { while (p < end && *p >= '0' && *p <= '9') ++p; }
    else return false;
    if (p < end && *p == '.') { ++p; if (p >= end || *p < '0' || *p > '9') return false; while (p < end && *p >= '0' && *p <= '9') ++p; }
    if (p < end && (*p == 'e' || *p == 'E')) {
      ++p;
      if (p < end && (*p == '+' || *p == '-')) ++p;
      if (p >= end || *p < '0' || *p > '9') return false;
      while (p < end && *p >= '0' && *p <= '9') ++p;
    }
    v.kind = J::Num;
    const auto r = std::from_chars(s, p, v.n);
    if (r.ec == std::errc::result_out_of_range) return false;      // (Go: "number out of range" for float64)
    return r.ec == std::errc();
  }
};

bool parse_object(const char* text, J& out) {
  if (!text) return false;
  Parser ps{text, text + std::strlen(text)};
  if (!ps.value(out)) return false;
  ps.ws();
  return ps.p == ps.end && out.kind == J::Obj;
}

void write_string(std::string& o, const std::string& s) {       // encoding/json encodeState.string with escapeHTML = true
  static const char* hex = "0123456789abcdef";
  o += '"';
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
      else if (c == '\n') o += "\\n";
      else if (c == '\r') o += "\\r";
      else if (c == '\t') o += "\\t";
      else if (c < 0x20 || c == '<' || c == '>' || c == '&') { o += "\\u00"; o += hex[c >> 4]; o += hex[c & 0xF]; }
      else o += (char)c;
      ++i;
      continue;
    }
    // decode one UTF-8 sequence; an invalid one is written as �
    int len = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : c >= 0xC0 ? 2 : 0;
    uint32_t cp = len == 4 ? (c & 7u) : len == 3 ? (c & 15u) : (c & 31u);
    bool good = len != 0 && i + (size_t)len <= s.size();
    for (int k = 1; good && k < len; ++k) {
      const unsigned char d = (unsigned char)s[i + k];
      if ((d & 0xC0) != 0x80) good = false;
      cp = (cp << 6) | (d & 0x3Fu);
    }
    if (good && ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp < 0xE000))) good = false;
    if (!good) { o += "\\ufffd"; ++i; continue; }
    if (cp == 0x2028 || cp == 0x2029) { o += "\\u202"; o += hex[cp & 0xF]; }
    else o.append(s, i, (size_t)len);
    i += (size_t)len;
  }
  o += '"';
}

void write_number(std::string& o, double f) {                    // encoding/json floatEncoder, 64 bit
  const double a = std::fabs(f);
  char buf[64];
  if (a != 0 && (a < 1e-6 || a >= 1e21)) {
    auto r = std::to_chars(buf, buf + sizeof buf, f, std::chars_format::scientific);
    std::string t(buf, r.ptr);
    const size_t n = t.size();
    if (n >= 4 && t[n - 4] == 'e' && t[n - 3] == '-' && t[n - 2] == '0') { t[n - 2] = t[n - 1]; t.pop_back(); }    // e-09 -> e-9
    o += t;
  } else {
    auto r = std::to_chars(buf, buf + sizeof buf, f, std::chars_format::fixed);
    o.append(buf, r.ptr);
  }
}

void write(std::string& o, const J& v) {
  switch (v.kind) {
    case J::Null: o += "null"; break;
    case J::Bool: o += v.b ? "true" : "false"; break;
    case J::Num: write_number(o, v.n); break;
    case J::Str: write_string(o, v.s); break;
    case J::Arr: {
      o += '[';
      for (size_t i = 0; i < v.a.size(); ++i) { if (i) o += ','; write(o, v.a[i]); }
      o += ']';
      break;
    }
    case J::Obj: {
      o += '{';
      bool first = true;
      for (const auto& kv : v.o) { if (!first) o += ','; first = false; write_string(o, kv.first); o += ':'; write(o, kv.second); }
      o += '}';
      break;
    }
  }
}

bool matches(const J& a, const J& b) {                           // merge.go matchesValue / matchesArray (P1)
  if (a.kind != b.kind) return false;
  switch (a.kind) {
    case J::Null: return true;
    case J::Bool: return a.b == b.b;
    case J::Num: return a.n == b.n;
    case J::Str: return a.s == b.s;
    case J::Arr:
      if (a.a.size() != b.a.size()) return false;
      for (size_t i = 0; i < a.a.size(); ++i) if (!matches(a.a[i], b.a[i])) return false;
      return true;
    case J::Obj: {
      static const J nil;
      for (const auto& kv : a.o) { auto it = b.o.find(kv.first); if (!matches(kv.second, it == b.o.end() ? nil : it->second)) return false; }
      for (const auto& kv : b.o) { auto it = a.o.find(kv.first); if (!matches(it == a.o.end() ? nil : it->second, kv.second)) return false; }
      return true;
    }
  }
  return false;
}

J diff(const J& a, const J& b) {                                  // merge.go getDiff (P1)
  J into;
  into.kind = J::Obj;
  for (const auto& kv : b.o) {
    const auto it = a.o.find(kv.first);
    if (it == a.o.end()) { into.o[kv.first] = kv.second; continue; }            // value was added
    const J &av = it->second, &bv = kv.second;
    if (av.kind != bv.kind) { into.o[kv.first] = bv; continue; }                // types have changed: replace completely
    switch (av.kind) {
      case J::Obj: { J d = diff(av, bv); if (!d.o.empty()) into.o[kv.first] = std::move(d); break; }
      case J::Str: case J::Num: case J::Bool: if (!matches(av, bv)) into.o[kv.first] = bv; break;
      case J::Arr: if (!matches(av, bv)) into.o[kv.first] = bv; break;
      case J::Null: break;                                                    // both null
    }
  }
  for (const auto& kv : a.o)                                                    // deleted values become null
    if (b.o.find(kv.first) == b.o.end()) into.o[kv.first] = J();
  return into;
}

int emit(const std::string& text, char* out, size_t cap, size_t* need) {
  if (need) *need = text.size() + 1;
  if (!out) return BS_OK;
  if (cap < text.size() + 1) return BS_ERR_CAPACITY;
  std::memcpy(out, text.c_str(), text.size() + 1);
  return BS_OK;
}

// ------------------------------------------------------------------------------------------------ PodGroupStatus
const char* const kPhaseNames[] = {"", "Pending", "Running", "PreScheduling", "Scheduling", "Scheduled", "Unknown", "Finished", "Failed"};   // types.go:28-56

std::string rfc3339(int64_t ns) {                                 // metav1.Time.MarshalJSON: UTC, seconds (P3)
  int64_t secs = ns / 1000000000;
  if (ns % 1000000000 < 0) --secs;
  int64_t days = secs / 86400, rem = secs % 86400;
  if (rem < 0) { rem += 86400; --days; }
  // civil date from days since 1970-01-01 (proleptic Gregorian)
  const int64_t z = days + 719468, era = (z >= 0 ? z : z - 146096) / 146097;
  const unsigned doe = (unsigned)(z - era * 146097), yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const int64_t y = (int64_t)yoe + era * 400;
  const unsigned doy = doe - (365 * yoe + yoe / 4 - yoe / 100), mp = (5 * doy + 2) / 153, d = doy - (153 * mp + 2) / 5 + 1, m = mp < 10 ? mp + 3 : mp - 9;
  char buf[48];
  std::snprintf(buf, sizeof buf, "%04lld-%02u-%02uT%02d:%02d:%02dZ", (long long)(y + (m <= 2)), m, d, (int)(rem / 3600), (int)(rem / 60 % 60), (int)(rem % 60));
  return buf;
}

std::string status_json(const bsh_pg_status& st, const char* occupied_by) {
  std::string o = "{\"phase\":";
  write_string(o, st.phase < 9 ? kPhaseNames[st.phase] : "");
  if (occupied_by && *occupied_by) { o += ",\"occupiedBy\":"; write_string(o, occupied_by); }      // `json:"occupiedBy,omitempty"`
  o += ",\"scheduled\":" + std::to_string(st.scheduled) + ",\"running\":" + std::to_string(st.running) + ",\"succeeded\":" + std::to_string(st.succeeded) +
       ",\"failed\":" + std::to_string(st.failed) + ",\"scheduleStartTime\":";
  if (st.schedule_start_ns == 0) o += "null";
  else write_string(o, rfc3339(st.schedule_start_ns));
  o += '}';
  return o;
}

bool same(const bsh_pg_status& a, const bsh_pg_status& b) {       // reflect.DeepEqual of the two objects: only Status can differ here
  return a.phase == b.phase && a.scheduled == b.scheduled && a.running == b.running && a.succeeded == b.succeeded && a.failed == b.failed &&
         a.occupied_by == b.occupied_by && a.schedule_start_ns == b.schedule_start_ns;
}

constexpr int64_t k48h = 48ll * 3600 * 1000000000;

}  // namespace

struct bsh_pg {
  std::set<uint64_t> succeed, failed;      // PodGroupMatchStatus.Succeed / .Failed (cache.go:52-67): uid sets that only grow
};

extern "C" {

bsh_pg* bsh_pg_new(void) { return new bsh_pg(); }
void bsh_pg_free(bsh_pg* pg) { delete pg; }
uint32_t bsh_pg_succeeded(const bsh_pg* pg) { return pg ? (uint32_t)pg->succeed.size() : 0u; }
uint32_t bsh_pg_failed(const bsh_pg* pg) { return pg ? (uint32_t)pg->failed.size() : 0u; }

const char* bsh_phase_name(uint32_t phase) { return phase < 9 ? kPhaseNames[phase] : ""; }
int bsh_phase_parse(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < 9; ++i)
    if (std::strcmp(name, kPhaseNames[i]) == 0) return i;
  return -1;
}
int bsh_phase_closed(uint32_t phase) {      // batchscheduler.go:258-261: anything but PreScheduling / Scheduling returns — Pending cannot occur there (Permit has run, core.go:279-281)
  return (phase != BSH_PHASE_PENDING && phase != BSH_PHASE_NONE && phase != BSH_PHASE_PRESCHEDULING && phase != BSH_PHASE_SCHEDULING) ? 1 : 0;
}

uint32_t bsh_pg_permit(uint32_t phase) { return phase == BSH_PHASE_PENDING ? (uint32_t)BSH_PHASE_PRESCHEDULING : phase; }      // core.go:279-281

int bsh_pg_post_bind(uint32_t min_member, const bsh_pg_status* in, int64_t now_ns, bsh_pg_status* out, uint8_t* patch) {
  if (!in || !out) return BS_ERR_INVALID;
  bsh_pg_status st = *in;
  st.scheduled = in->scheduled + 1u;                                            // core.go:327
  if (st.scheduled >= min_member) st.phase = BSH_PHASE_SCHEDULED;               // :329-330
  else {
    st.phase = BSH_PHASE_SCHEDULING;                                            // :331-332
    if (st.schedule_start_ns == 0) st.schedule_start_ns = now_ns;               // :333-335
  }
  if (patch) *patch = st.phase != in->phase ? 1 : 0;                            // :338
  *out = st;
  return BS_OK;
}

int bsh_pg_start_gate(uint32_t min_member, const bsh_pg_status* in, uint8_t* release, uint8_t* stamp) {
  if (!in) return BS_ERR_INVALID;
  const bool open = in->phase == BSH_PHASE_PRESCHEDULING || in->phase == BSH_PHASE_SCHEDULING;    // batchscheduler.go:258-261
  if (release) *release = open ? 1 : 0;
  if (stamp) *stamp = (open && in->scheduled >= min_member) ? 1 : 0;                                // :264-285
  return BS_OK;
}

int bsh_pg_enqueue(uint32_t min_member, int64_t creation_ns, const bsh_pg_status* st) {      // pgAdded, controller.go:111-130
  if (!st) return 0;
  if (st->phase == BSH_PHASE_FINISHED || st->phase == BSH_PHASE_FAILED) return 0;                                                   // :118-120
  if (st->scheduled == min_member && st->running == 0 && st->schedule_start_ns != 0 && st->schedule_start_ns - creation_ns > k48h) return 0;   // :122-125
  return 1;
}

int bsh_pg_sync(bsh_pg* pg, uint32_t min_member, int64_t creation_ns, const bsh_pg_status* in, const uint64_t* pod_uids, const uint8_t* pod_phases,
                uint32_t npods, bsh_pg_status* recovered, bsh_pg_status* out, uint32_t* actions) {
  if (!pg || !in || !out || !actions || (npods && (!pod_uids || !pod_phases))) return BS_ERR_INVALID;
  bsh_pg_status base = *in, st = *in;          // base: the object the server holds (pg), st: pgCopy
  uint32_t act = 0;
  if (st.phase == BSH_PHASE_NONE) st.phase = BSH_PHASE_PENDING;                                   // controller.go:199-200
  else if (st.phase == BSH_PHASE_PENDING && st.schedule_start_ns != 0) {                          // :201-223 recover from abnormal exit
    act |= BSH_SYNC_LISTED_PODS;
    st.scheduled = npods;                                                                        // :210
    if (st.scheduled > 0 && !same(base, st)) {                                                   // :211
      act |= BSH_SYNC_PATCH_RECOVER;                                                             // :212-220
      base = st;                                                                                 // pg = the patched object
      if (recovered) *recovered = st;
    }
  }
  // :224-226: the cache entry takes pgCopy.Status (the caller's bsh_sop / Go cache)
  if (st.scheduled == min_member && st.running == 0 && st.schedule_start_ns != 0 && st.schedule_start_ns - creation_ns > k48h) {      // :227-231
    *out = st;                                 // (the zero time lies 2000 years before any creation stamp: Sub saturates negative)
    *actions = act | BSH_SYNC_NO_REQUEUE;
    return BS_OK;
  }
  if (st.phase == BSH_PHASE_SCHEDULED || st.phase == BSH_PHASE_RUNNING || st.phase == BSH_PHASE_SCHEDULING) {      // :235-236
    act |= BSH_SYNC_LISTED_PODS;
    uint32_t not_pending = 0, running = 0;
    for (uint32_t i = 0; i < npods; ++i) {                                                       // :248-262
      switch (pod_phases[i]) {
        case BSH_POD_RUNNING: running++; break;
        case BSH_POD_SUCCEEDED: pg->succeed.insert(pod_uids[i]); break;
        case BSH_POD_FAILED: pg->failed.insert(pod_uids[i]); break;
        default: break;
      }
      if (pod_phases[i] != BSH_POD_PENDING) not_pending++;
    }
    st.failed = (uint32_t)pg->failed.size();                                                     // :265
    st.succeeded = (uint32_t)pg->succeed.size();                                                 // :266
    st.running = running;                                                                        // :267
    if (not_pending > st.scheduled) st.scheduled = not_pending;                                  // :269-272
    if (not_pending < min_member && not_pending != 0) {                                          // :275-279 recover from exit
      st.scheduled = not_pending;
      st.phase = BSH_PHASE_SCHEDULING;
    }
    if ((uint32_t)(st.succeeded + st.running) >= min_member) st.phase = BSH_PHASE_RUNNING;       // :281-283
    if (st.failed != 0 && (uint32_t)(st.failed + st.running + st.succeeded) >= min_member) st.phase = BSH_PHASE_FAILED;      // :284-288
    if (st.succeeded >= min_member) st.phase = BSH_PHASE_FINISHED;                               // :289-291
  }
  if (!same(base, st)) {                                                                         // :293
    act |= BSH_SYNC_PATCH;
    if (st.phase == BSH_PHASE_FINISHED || st.phase == BSH_PHASE_FAILED) act |= BSH_SYNC_CACHE_DELETE;      // :304-306
  }
  *out = st;
  *actions = act;
  return BS_OK;
}

int bsh_merge_patch(const char* original, const char* modified, char* out, size_t cap, size_t* need) {
  J a, b;
  if (!parse_object(original, a) || !parse_object(modified, b)) return BS_ERR_INVALID;
  std::string text;
  write(text, diff(a, b));
  return emit(text, out, cap, need);
}

int bsh_pg_status_json(const bsh_pg_status* st, const char* occupied_by, char* out, size_t cap, size_t* need) {
  if (!st) return BS_ERR_INVALID;
  return emit(status_json(*st, occupied_by), out, cap, need);
}

int bsh_pg_status_patch(const bsh_pg_status* from, const bsh_pg_status* to, const char* occupied_by, char* out, size_t cap, size_t* need) {
  if (!from || !to) return BS_ERR_INVALID;
  const std::string a = "{\"status\":" + status_json(*from, occupied_by) + "}", b = "{\"status\":" + status_json(*to, occupied_by) + "}";
  return bsh_merge_patch(a.c_str(), b.c_str(), out, cap, need);
}

}  // extern "C"
