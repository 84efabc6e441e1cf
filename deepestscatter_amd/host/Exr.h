// Exr.h -- a minimal OpenEXR 2 writer for what Camera::saveToDisk produces (Camera.cpp:149-175): one part,
// scan lines, channels B, G, R of type FLOAT, data window = display window = the frame, line order
// DECREASING_Y, pixel (x, y) = progressive[y * width + x].  OpenEXR itself is not on the target image; the
// file is written uncompressed (the reference's Header defaults to ZIP; every EXR reader takes both).
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace DeepestScatter
{
    namespace Exr
    {
        inline void put(std::vector<char>& b, const void* p, size_t n) { b.insert(b.end(), (const char*)p, (const char*)p + n); }
        inline void putString(std::vector<char>& b, const char* s) { put(b, s, std::strlen(s) + 1); }
        template <class T> inline void putValue(std::vector<char>& b, T v) { put(b, &v, sizeof v); }
        inline void attribute(std::vector<char>& b, const char* name, const char* type, const std::vector<char>& value)
        {
            putString(b, name);
            putString(b, type);
            putValue<int32_t>(b, (int32_t)value.size());
            put(b, value.data(), value.size());
        }

        // rgba: width*height float4 (alpha ignored), row 0 first
        inline void writeRgbFloat(const std::string& path, uint32_t width, uint32_t height, const float* rgba)
        {
            std::vector<char> h;
            const unsigned char magic[8] = { 0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0 }; // magic, version 2, no flags
            put(h, magic, 8);
            {
                std::vector<char> v;
                for (const char* name : { "B", "G", "R" }) // alphabetical, as the library stores them
                {
                    putString(v, name);
                    putValue<int32_t>(v, 2); // FLOAT
                    putValue<uint8_t>(v, 0); // pLinear
                    put(v, "\0\0\0", 3);
                    putValue<int32_t>(v, 1); // xSampling
                    putValue<int32_t>(v, 1); // ySampling
                }
                putValue<uint8_t>(v, 0);
                attribute(h, "channels", "chlist", v);
            }
            { std::vector<char> v; putValue<uint8_t>(v, 0); attribute(h, "compression", "compression", v); } // NO_COMPRESSION
            for (const char* name : { "dataWindow", "displayWindow" })
            {
                std::vector<char> v;
                putValue<int32_t>(v, 0); putValue<int32_t>(v, 0);
                putValue<int32_t>(v, (int32_t)width - 1); putValue<int32_t>(v, (int32_t)height - 1);
                attribute(h, name, "box2i", v);
            }
            { std::vector<char> v; putValue<uint8_t>(v, 1); attribute(h, "lineOrder", "lineOrder", v); }    // DECREASING_Y
            { std::vector<char> v; putValue<float>(v, 1.0f); attribute(h, "pixelAspectRatio", "float", v); }
            { std::vector<char> v; putValue<float>(v, 0.0f); putValue<float>(v, 0.0f); attribute(h, "screenWindowCenter", "v2f", v); }
            { std::vector<char> v; putValue<float>(v, 1.0f); attribute(h, "screenWindowWidth", "float", v); }
            putValue<uint8_t>(h, 0); // end of header

            const uint64_t rowBytes = 3ull * width * sizeof(float);
            const uint64_t chunkBytes = 8 + rowBytes; // y, size, data
            const uint64_t dataStart = h.size() + 8ull * height;
            // the offset table is indexed by y - minY whatever the line order; the chunks themselves are
            // stored from the last scan line to the first (DECREASING_Y)
            for (uint32_t y = 0; y < height; y++) putValue<uint64_t>(h, dataStart + (uint64_t)(height - 1 - y) * chunkBytes);
            std::ofstream f(path, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + path);
            f.write(h.data(), (std::streamsize)h.size());
            std::vector<float> row(3 * (size_t)width);
            for (uint32_t k = 0; k < height; k++)
            {
                const uint32_t y = height - 1 - k;
                const int32_t yy = (int32_t)y, size = (int32_t)rowBytes;
                for (uint32_t c = 0; c < 3; c++) // B, G, R planes of the scan line
                    for (uint32_t x = 0; x < width; x++) row[(size_t)c * width + x] = rgba[((size_t)y * width + x) * 4 + (2 - c)];
                f.write((const char*)&yy, 4);
                f.write((const char*)&size, 4);
                f.write((const char*)row.data(), (std::streamsize)rowBytes);
            }
            if (!f) throw std::runtime_error("write error on " + path);
        }
    }
}
