/* TEST INFRASTRUCTURE (oracle/seam) — src/util.h:19 includes this header but its declarations use nothing from it (the
 * text-format reader lives in util.cc, which the seam does not compile). */
#pragma once
