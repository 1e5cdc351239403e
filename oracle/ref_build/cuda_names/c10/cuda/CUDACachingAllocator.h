// TEST INFRASTRUCTURE ONLY: empty stand-in so core/memory/memory_pool.h parses (see ../../cuda_runtime_api.h)
#pragma once
