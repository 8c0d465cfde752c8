// kaldi-io.cc -- see kaldi-io.h.
#include "kaldi-io.h"

#include <ext/stdio_filebuf.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>

namespace klat {

int g_verbose = 0;

void LogMessage(const char* level, const std::string& msg) {
  std::cerr << level << " (klat) " << msg << std::endl;
}

Specifier ParseSpecifier(const std::string& spec, bool for_writing) {
  Specifier s;
  const size_t colon = spec.find(':');
  if (colon == std::string::npos) KLAT_ERR("Invalid " << (for_writing ? "wspecifier" : "rspecifier") << " \"" << spec << "\"");
  std::stringstream opts(spec.substr(0, colon));
  std::string o;
  while (std::getline(opts, o, ',')) {
    if (o == "ark") s.is_ark = true;
    else if (o == "t") s.text = true;
    else if (o == "b") s.text = false;
    else if (o == "scp") KLAT_ERR("scp: tables are not supported by this build (use ark:)");
    else if (o == "f" || o == "nf" || o == "p" || o == "np" || o == "o" || o == "no" || o == "s" || o == "ns" || o == "cs" || o == "ncs" || o == "bg") {
      // permissive / flush / sorted hints: no effect on a sequential archive reader or writer
    } else {
      KLAT_ERR("Unknown option \"" << o << "\" in specifier \"" << spec << "\"");
    }
  }
  if (!s.is_ark) KLAT_ERR("Only ark: specifiers are supported, got \"" << spec << "\"");
  s.filename = spec.substr(colon + 1);
  while (!s.filename.empty() && s.filename.front() == ' ') s.filename.erase(0, 1);
  while (!s.filename.empty() && s.filename.back() == ' ') s.filename.pop_back();
  if (s.filename.empty()) KLAT_ERR("Empty filename in specifier \"" << spec << "\"");
  return s;
}

Input::Input(const std::string& filename) {
  if (filename == "-") {
    is_ = &std::cin;
  } else if (!filename.empty() && filename.back() == '|') {
    const std::string cmd = filename.substr(0, filename.size() - 1);
    pipe_ = popen(cmd.c_str(), "r");
    if (!pipe_) KLAT_ERR("Failed opening pipe for reading, command is: " << cmd);
    buf_.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_, std::ios_base::in | std::ios_base::binary));
    owned_.reset(new std::istream(buf_.get()));
    is_ = owned_.get();
  } else {
    std::unique_ptr<std::ifstream> f(new std::ifstream());
    iobuf_.resize((size_t)1 << 20);
    f->rdbuf()->pubsetbuf(iobuf_.data(), (std::streamsize)iobuf_.size());
    f->open(filename, std::ios_base::in | std::ios_base::binary);
    if (!f->is_open()) KLAT_ERR("Error opening input stream " << filename);
    owned_ = std::move(f);
    is_ = owned_.get();
  }
}
Input::~Input() {
  owned_.reset();
  buf_.reset();
  if (pipe_) pclose(pipe_);
}

Output::Output(const std::string& filename) {
  if (filename == "-") {
    os_ = &std::cout;
  } else if (!filename.empty() && filename.front() == '|') {
    const std::string cmd = filename.substr(1);
    pipe_ = popen(cmd.c_str(), "w");
    if (!pipe_) KLAT_ERR("Failed opening pipe for writing, command is: " << cmd);
    buf_.reset(new __gnu_cxx::stdio_filebuf<char>(pipe_, std::ios_base::out | std::ios_base::binary));
    owned_.reset(new std::ostream(buf_.get()));
    os_ = owned_.get();
  } else {
    std::unique_ptr<std::ofstream> f(new std::ofstream());
    iobuf_.resize((size_t)1 << 20);
    f->rdbuf()->pubsetbuf(iobuf_.data(), (std::streamsize)iobuf_.size());
    f->open(filename, std::ios_base::out | std::ios_base::binary);
    if (!f->is_open()) KLAT_ERR("Error opening output stream " << filename);
    owned_ = std::move(f);
    os_ = owned_.get();
  }
}
void Output::Close() {
  if (os_) os_->flush();
  owned_.reset();
  buf_.reset();
  if (pipe_) {
    pclose(pipe_);
    pipe_ = nullptr;
  }
  os_ = nullptr;
}
Output::~Output() { Close(); }

void InitKaldiOutputStream(std::ostream& os, bool binary) {
  if (binary) {
    os.put('\0');
    os.put('B');
  }
  if (os.precision() < 7) os.precision(7);
}
template <class T>
static void write_bin(std::ostream& os, T v) {
  os.put((char)sizeof(T));
  os.write(reinterpret_cast<const char*>(&v), sizeof(T));
}
void WriteBasicType(std::ostream& os, bool binary, int32_t v) {
  if (binary) write_bin(os, v);
  else os << v << " ";
}
void WriteBasicType(std::ostream& os, bool binary, float v) {
  if (binary) write_bin(os, v);
  else os << v << " ";
}
void WriteBasicType(std::ostream& os, bool binary, double v) {
  if (binary) write_bin(os, v);
  else os << v << " ";
}
void WriteToken(std::ostream& os, bool, const std::string& tok) { os << tok << " "; }

// ---- ParseOptions ----
static std::string norm(const std::string& n) {
  std::string r = n;
  for (auto& c : r)
    if (c == '_') c = '-';
  return r;
}
void ParseOptions::Register(const std::string& name, float* v, const std::string& doc) {
  std::ostringstream d;
  d << *v;
  opts_[norm(name)] = Opt{'f', v, doc, d.str()};
}
void ParseOptions::Register(const std::string& name, int32_t* v, const std::string& doc) {
  opts_[norm(name)] = Opt{'i', v, doc, std::to_string(*v)};
}
void ParseOptions::Register(const std::string& name, bool* v, const std::string& doc) {
  opts_[norm(name)] = Opt{'b', v, doc, *v ? "true" : "false"};
}
void ParseOptions::Register(const std::string& name, std::string* v, const std::string& doc) {
  opts_[norm(name)] = Opt{'s', v, doc, *v};
}
void ParseOptions::Set(const std::string& name_in, const std::string& value, bool has_value) {
  const std::string name = norm(name_in);
  if (name == "help") {
    PrintUsage();
    std::exit(0);
  }
  if (name == "verbose") {
    g_verbose = std::atoi(value.c_str());
    return;
  }
  if (name == "print-args") {
    print_args_ = !(value == "false" || value == "0");
    return;
  }
  if (name == "config") {
    ReadConfig(value);
    return;
  }
  auto it = opts_.find(name);
  if (it == opts_.end()) KLAT_ERR("Invalid option --" << name_in);
  Opt& o = it->second;
  char* end = nullptr;
  switch (o.type) {
    case 'f': {
      if (value == "inf" || value == "+inf" || value == "Infinity" || value == "infinity") {
        *static_cast<float*>(o.ptr) = std::numeric_limits<float>::infinity();
        break;
      }
      const double d = std::strtod(value.c_str(), &end);
      if (!has_value || end == value.c_str() || *end) KLAT_ERR("Invalid floating-point option --" << name_in << "=" << value);
      *static_cast<float*>(o.ptr) = (float)d;
      break;
    }
    case 'i': {
      const long l = std::strtol(value.c_str(), &end, 10);
      if (!has_value || end == value.c_str() || *end) KLAT_ERR("Invalid integer option --" << name_in << "=" << value);
      *static_cast<int32_t*>(o.ptr) = (int32_t)l;
      break;
    }
    case 'b':
      if (!has_value || value == "true" || value == "t" || value == "1") *static_cast<bool*>(o.ptr) = true;
      else if (value == "false" || value == "f" || value == "0") *static_cast<bool*>(o.ptr) = false;
      else KLAT_ERR("Invalid boolean option --" << name_in << "=" << value);
      break;
    case 's':
      *static_cast<std::string*>(o.ptr) = value;
      break;
  }
}
void ParseOptions::ReadConfig(const std::string& file) {
  std::ifstream f(file);
  if (!f.is_open()) KLAT_ERR("Cannot open config file " << file);
  std::string line;
  while (std::getline(f, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    while (!line.empty() && isspace((unsigned char)line.back())) line.pop_back();
    size_t b = 0;
    while (b < line.size() && isspace((unsigned char)line[b])) ++b;
    line.erase(0, b);
    if (line.empty()) continue;
    if (line.compare(0, 2, "--") != 0) KLAT_ERR("Bad line in config file " << file << ": " << line);
    const size_t eq = line.find('=');
    if (eq == std::string::npos) Set(line.substr(2), "", false);
    else Set(line.substr(2, eq - 2), line.substr(eq + 1), true);
  }
}
int ParseOptions::Read(int argc, const char* const* argv) {
  for (int i = 0; i < argc; ++i) cmdline_ += std::string(i ? " " : "") + argv[i];
  int i = 1;
  for (; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--") {
      ++i;
      break;
    }
    if (a.compare(0, 2, "--") != 0) break;
    const size_t eq = a.find('=');
    if (eq == std::string::npos) Set(a.substr(2), "", false);
    else Set(a.substr(2, eq - 2), a.substr(eq + 1), true);
  }
  for (; i < argc; ++i) args_.push_back(argv[i]);
  if (print_args_) std::cerr << cmdline_ << std::endl;
  return 0;
}
const std::string& ParseOptions::GetArg(int i) const {
  if (i < 1 || i > (int)args_.size()) KLAT_ERR("ParseOptions::GetArg, invalid index " << i);
  return args_[i - 1];
}
void ParseOptions::PrintUsage(bool print_command_line) const {
  std::cerr << "\n" << usage_ << "\nOptions:\n";
  for (const auto& kv : opts_)
    std::cerr << "  --" << kv.first << " : " << kv.second.doc << " (default = " << kv.second.def << ")\n";
  std::cerr << "\nStandard options:\n  --config, --help, --print-args, --verbose\n\n";
  if (print_command_line) std::cerr << "Command line was: " << cmdline_ << "\n";
}

}  // namespace klat
