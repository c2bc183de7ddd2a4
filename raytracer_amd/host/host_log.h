// Minimal levelled logger for the host side: `[HH:MM:SS] [LEVEL] message`, Error and above to
// stderr, level chosen by --log-level (the reference's Utils::Logger, Code/Utils/logger.h:33-156,
// reduced to what the command line needs).
#pragma once

#include <cstdarg>
#include <cstdio>
#include <ctime>
#include <string>

enum class LogLevel { Debug = 0, Info = 1, Warn = 2, Error = 3, Fatal = 4 };

inline LogLevel &host_log_threshold() {
  static LogLevel level = LogLevel::Info;
  return level;
}

inline void host_log_set_level(const std::string &name) {
  if (name == "debug") host_log_threshold() = LogLevel::Debug;
  else if (name == "info") host_log_threshold() = LogLevel::Info;
  else if (name == "warn") host_log_threshold() = LogLevel::Warn;
  else if (name == "error") host_log_threshold() = LogLevel::Error;
  else if (name == "fatal") host_log_threshold() = LogLevel::Fatal;
}

inline void host_log(LogLevel level, const char *fmt, ...) {
  if (level < host_log_threshold()) return;
  static const char *names[] = {"DEBUG", "INFO", "WARN", "ERROR", "FATAL"};
  std::FILE *to = level >= LogLevel::Error ? stderr : stdout;
  char stamp[16];
  const std::time_t now = std::time(nullptr);
  std::tm tmv;
  localtime_r(&now, &tmv);
  std::strftime(stamp, sizeof stamp, "%H:%M:%S", &tmv);
  std::fprintf(to, "[%s] [%s] ", stamp, names[static_cast<int>(level)]);
  va_list ap;
  va_start(ap, fmt);
  std::vfprintf(to, fmt, ap);
  va_end(ap);
  std::fputc('\n', to);
}
