#pragma once
/* Export macro kept for source compatibility with the reference's headers (ML/dll.hpp:8-20). */
#ifndef DLL_DECLSPEC
#define DLL_DECLSPEC __attribute__((visibility("default")))
#endif
