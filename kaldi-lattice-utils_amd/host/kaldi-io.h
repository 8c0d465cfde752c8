// kaldi-io.h -- the slice of Kaldi's table I/O the three tools need:
// rspecifier / wspecifier parsing ("ark:file", "ark,t:-", "ark:zcat f.gz|", "ark:|gzip > f"),
// text/binary basic types, and Kaldi-style options parsing.  Written from the wire formats, not
// from Kaldi sources (which are not in the reference tree).
#ifndef KLAT_HOST_KALDI_IO_H_
#define KLAT_HOST_KALDI_IO_H_
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace klat {

// ---- logging in the style of KALDI_LOG / KALDI_WARN / KALDI_ERR (stderr; ERR throws) ----
extern int g_verbose;
void LogMessage(const char* level, const std::string& msg);
#define KLAT_LOG(msg)                          \
  do {                                         \
    std::ostringstream _os;                    \
    _os << msg;                                \
    ::klat::LogMessage("LOG", _os.str());      \
  } while (0)
#define KLAT_VLOG(v, msg)                      \
  do {                                         \
    if (::klat::g_verbose >= (v)) KLAT_LOG(msg); \
  } while (0)
#define KLAT_WARN(msg)                         \
  do {                                         \
    std::ostringstream _os;                    \
    _os << msg;                                \
    ::klat::LogMessage("WARNING", _os.str());  \
  } while (0)
#define KLAT_LOG(msg)                          \
  do {                                         \
    std::ostringstream _os;                    \
    _os << msg;                                \
    ::klat::LogMessage("LOG", _os.str());      \
  } while (0)
#define KLAT_ERR(msg)                          \
  do {                                         \
    std::ostringstream _os;                    \
    _os << msg;                                \
    ::klat::LogMessage("ERROR", _os.str());    \
    throw std::runtime_error(_os.str());       \
  } while (0)

// ---- specifiers ----
struct Specifier {
  bool is_ark = false;
  bool text = false;      // ",t" option (writing)
  std::string filename;   // "-", a path, "cmd|" (read pipe) or "|cmd" (write pipe)
};
Specifier ParseSpecifier(const std::string& spec, bool for_writing);

// std::istream / std::ostream over a file, stdin/stdout or a popen()ed command.
class Input {
 public:
  explicit Input(const std::string& filename);
  ~Input();
  std::istream& Stream() { return *is_; }

 private:
  std::unique_ptr<std::istream> owned_;
  std::istream* is_ = nullptr;
  FILE* pipe_ = nullptr;
  std::unique_ptr<std::streambuf> buf_;
  std::vector<char> iobuf_;  // 1 MiB stream buffer of a file (set before it is opened)
};
class Output {
 public:
  explicit Output(const std::string& filename);
  ~Output();
  std::ostream& Stream() { return *os_; }
  void Close();

 private:
  std::unique_ptr<std::ostream> owned_;
  std::ostream* os_ = nullptr;
  FILE* pipe_ = nullptr;
  std::unique_ptr<std::streambuf> buf_;
  std::vector<char> iobuf_;
};

// ---- basic types (Kaldi WriteBasicType: binary = 1-byte size tag + little-endian value;
// text = value followed by a space, floats at 7 significant digits) ----
void InitKaldiOutputStream(std::ostream& os, bool binary);  // "\0B" in binary mode, precision 7 in text
void WriteBasicType(std::ostream& os, bool binary, int32_t v);
void WriteBasicType(std::ostream& os, bool binary, float v);
void WriteBasicType(std::ostream& os, bool binary, double v);
void WriteToken(std::ostream& os, bool binary, const std::string& tok);

// ---- ParseOptions look-alike: --name=value, --flag, --config=file, --help, --verbose,
// --print-args; positional arguments afterwards ----
class ParseOptions {
 public:
  explicit ParseOptions(const std::string& usage) : usage_(usage) {}
  void Register(const std::string& name, float* v, const std::string& doc);
  void Register(const std::string& name, int32_t* v, const std::string& doc);
  void Register(const std::string& name, bool* v, const std::string& doc);
  void Register(const std::string& name, std::string* v, const std::string& doc);
  // returns 0; on --help prints usage and exits(0); on a bad option throws
  int Read(int argc, const char* const* argv);
  int NumArgs() const { return (int)args_.size(); }
  const std::string& GetArg(int i) const;  // 1-based like Kaldi
  void PrintUsage(bool print_command_line = false) const;

 private:
  struct Opt {
    char type;
    void* ptr;
    std::string doc, def;
  };
  void Set(const std::string& name, const std::string& value, bool has_value);
  void ReadConfig(const std::string& file);
  std::string usage_, cmdline_;
  std::map<std::string, Opt> opts_;
  std::vector<std::string> args_;
  bool print_args_ = true;
};

}  // namespace klat
#endif
