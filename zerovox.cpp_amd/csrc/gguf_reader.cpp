// gguf_reader.cpp — GGUF v3 parser (see gguf_reader.h for what it replaces).
#include "gguf_reader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>

#include "common.h"

namespace zv
{

namespace
{

// gguf_type codes (reference ggml/include/ggml.h, enum gguf_type)
enum : uint32_t
{
    T_U8 = 0, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64
};

size_t scalar_size(uint32_t t)
{
    switch (t)
    {
        case T_U8: case T_I8: case T_BOOL: return 1;
        case T_U16: case T_I16: return 2;
        case T_U32: case T_I32: case T_F32: return 4;
        case T_U64: case T_I64: case T_F64: return 8;
        default: return 0;
    }
}

struct Cursor
{
    const uint8_t *p;
    size_t         n, pos = 0;
    const char    *path;

    void need(size_t k) const
    {
        if (pos + k > n) fail(ZV_ERR_FORMAT, "%s: truncated GGUF file (need %zu bytes at offset %zu)", path, k, pos);
    }
    template <typename T> T rd()
    {
        need(sizeof(T));
        T v;
        memcpy(&v, p + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
    std::string str()
    {
        uint64_t len = rd<uint64_t>();
        if (len > (1u << 20)) fail(ZV_ERR_FORMAT, "%s: unreasonable string length %llu", path, (unsigned long long)len);
        need(len);
        std::string s((const char *)p + pos, (size_t)len);
        pos += len;
        return s;
    }
    void skip(size_t k) { need(k); pos += k; }
};

size_t type_size(uint32_t t)
{
    switch (t)
    {
        case GGML_F32: return 4;
        case GGML_F16: return 2;
        case GGML_I32: return 4;
        default: return 0;
    }
}

}  // namespace

GgufFile::~GgufFile()
{
    if (map_) munmap(map_, map_size_);
}

void GgufFile::open(const std::string &path)
{
    int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) fail(ZV_ERR_IO, "cannot open '%s'", path.c_str());
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 24)
    {
        ::close(fd);
        fail(ZV_ERR_FORMAT, "%s: too small to be a GGUF file", path.c_str());
    }
    map_size_ = (size_t)st.st_size;
    map_ = mmap(nullptr, map_size_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED)
    {
        map_ = nullptr;
        fail(ZV_ERR_IO, "mmap of '%s' failed", path.c_str());
    }

    Cursor c{(const uint8_t *)map_, map_size_, 0, path.c_str()};
    if (memcmp(c.p, "GGUF", 4) != 0) fail(ZV_ERR_FORMAT, "%s: bad magic (not a GGUF file)", path.c_str());
    c.pos = 4;
    version_ = c.rd<uint32_t>();
    if (version_ != 3) fail(ZV_ERR_FORMAT, "%s: GGUF version %u not supported (need 3)", path.c_str(), version_);
    const uint64_t n_tensors = c.rd<uint64_t>();
    const uint64_t n_kv = c.rd<uint64_t>();
    if (n_tensors > (1u << 20) || n_kv > (1u << 20)) fail(ZV_ERR_FORMAT, "%s: unreasonable tensor/KV count", path.c_str());

    size_t alignment = 32;                      // GGUF_DEFAULT_ALIGNMENT unless `general.alignment` says otherwise
    for (uint64_t i = 0; i < n_kv; i++)
    {
        std::string key = c.str();
        const uint32_t t = c.rd<uint32_t>();
        kv_type_[key] = t;
        if (t == T_STR)
            c.str();
        else if (t == T_ARR)
        {
            const uint32_t et = c.rd<uint32_t>();
            const uint64_t cnt = c.rd<uint64_t>();
            if (et == T_STR)
                for (uint64_t j = 0; j < cnt; j++) c.str();
            else
            {
                const size_t es = scalar_size(et);
                if (!es) fail(ZV_ERR_FORMAT, "%s: bad array element type %u for key %s", path.c_str(), et, key.c_str());
                c.skip(es * cnt);
            }
        }
        else if (t == T_U32)
        {
            const uint32_t v = c.rd<uint32_t>();
            kv_u32_[key] = v;
            if (key == "general.alignment") alignment = v;
        }
        else
        {
            const size_t es = scalar_size(t);
            if (!es) fail(ZV_ERR_FORMAT, "%s: bad KV type %u for key %s", path.c_str(), t, key.c_str());
            c.skip(es);
        }
    }
    if (alignment == 0 || (alignment & (alignment - 1))) fail(ZV_ERR_FORMAT, "%s: bad alignment %zu", path.c_str(), alignment);

    struct Info { uint64_t offset; };
    std::vector<Info> infos;
    tensors_.reserve(n_tensors);
    for (uint64_t i = 0; i < n_tensors; i++)
    {
        GgufTensor t;
        t.name = c.str();
        t.n_dims = c.rd<uint32_t>();
        if (t.n_dims < 1 || t.n_dims > 4) fail(ZV_ERR_FORMAT, "%s: tensor %s has %u dims", path.c_str(), t.name.c_str(), t.n_dims);
        for (uint32_t d = 0; d < t.n_dims; d++)
        {
            t.ne[d] = (int64_t)c.rd<uint64_t>();
            if (t.ne[d] <= 0 || t.ne[d] > (int64_t)1 << 32) fail(ZV_ERR_FORMAT, "%s: tensor %s has a bad extent", path.c_str(), t.name.c_str());
        }
        t.type = c.rd<uint32_t>();
        const uint64_t off = c.rd<uint64_t>();
        const size_t ts = type_size(t.type);
        if (!ts) fail(ZV_ERR_FORMAT, "%s: tensor %s has unsupported ggml type %u (only F32/F16/I32)", path.c_str(), t.name.c_str(), t.type);
        t.nbytes = (size_t)t.nelements() * ts;
        infos.push_back({off});
        tensors_.push_back(std::move(t));
    }
    const size_t data0 = (c.pos + alignment - 1) / alignment * alignment;
    for (size_t i = 0; i < tensors_.size(); i++)
    {
        GgufTensor &t = tensors_[i];
        const uint64_t off = infos[i].offset;
        if (off % alignment) fail(ZV_ERR_FORMAT, "%s: tensor %s is not aligned", path.c_str(), t.name.c_str());
        if (data0 + off + t.nbytes > map_size_) fail(ZV_ERR_FORMAT, "%s: tensor %s runs past the end of the file", path.c_str(), t.name.c_str());
        t.data = (const uint8_t *)map_ + data0 + off;
        index_[t.name] = i;
    }
}

bool GgufFile::has_u32(const std::string &key) const { return kv_u32_.count(key) != 0; }

uint32_t GgufFile::get_u32(const std::string &key) const
{
    auto it = kv_u32_.find(key);
    if (it != kv_u32_.end()) return it->second;
    if (kv_type_.count(key)) fail(ZV_ERR_FORMAT, "key %s has wrong type (need u32)", key.c_str());
    fail(ZV_ERR_MISSING, "key not found in model: %s", key.c_str());
}

const GgufTensor *GgufFile::find(const std::string &name) const
{
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors_[it->second];
}

const GgufTensor &GgufFile::get(const std::string &name) const
{
    const GgufTensor *t = find(name);
    if (!t) fail(ZV_ERR_MISSING, "tensor '%s' not found", name.c_str());
    return *t;
}

}  // namespace zv
