// gguf_reader.h — own GGUF v3 reader (the weight-file half of the drop-in boundary).
//
// Replaces the reference's use of ggml's gguf_init_from_file / gguf_find_key / gguf_get_val_u32 /
// gguf_get_tensor_offset (reference src/zerovox.cpp:28-56,140-172; format: ggml/src/ggml.c:6455-6476,
// 6620-6909).  The file is mmap'ed; tensors are views into the mapping.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace zv
{

enum GgmlType : uint32_t { GGML_F32 = 0, GGML_F16 = 1, GGML_I32 = 26 };

struct GgufTensor
{
    std::string name;
    uint32_t    n_dims = 0;
    int64_t     ne[4] = {1, 1, 1, 1};     // ggml order: ne[0] fastest; missing dims padded with 1
    uint32_t    type = 0;
    const void *data = nullptr;
    size_t      nbytes = 0;
    int64_t     nelements() const { return ne[0] * ne[1] * ne[2] * ne[3]; }
};

class GgufFile
{
  public:
    GgufFile() = default;
    ~GgufFile();
    GgufFile(const GgufFile &) = delete;
    GgufFile &operator=(const GgufFile &) = delete;

    // throws zv::Error (status + message) on any problem
    void open(const std::string &path);

    bool              has_u32(const std::string &key) const;
    uint32_t          get_u32(const std::string &key) const;          // ZV_ERR_MISSING / ZV_ERR_FORMAT (wrong type)
    const GgufTensor *find(const std::string &name) const;
    const GgufTensor &get(const std::string &name) const;             // ZV_ERR_MISSING
    const std::vector<GgufTensor> &tensors() const { return tensors_; }
    uint32_t          version() const { return version_; }

  private:
    void  *map_ = nullptr;
    size_t map_size_ = 0;
    uint32_t version_ = 0;
    std::map<std::string, uint32_t> kv_type_;
    std::map<std::string, uint32_t> kv_u32_;
    std::vector<GgufTensor> tensors_;
    std::map<std::string, size_t> index_;
};

}  // namespace zv
